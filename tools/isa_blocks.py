#!/usr/bin/env python3
"""Static instruction mix of one kernel of a HIP source, per basic block (no GPU needed).

    python tools/isa_blocks.py metamdbg_amd/csrc/scan.hip _ZN4mdbg11scan_kernelILb1ELb0ELb0EEEvNS_8ScanArgsE [min_instrs]

Compiles the file for gfx950 to assembly (hipcc --cuda-device-only -S), cuts the named kernel into basic blocks and
counts multiplies / other VALU / SALU / LDS / memory / waits in each block of at least `min_instrs` instructions.
Together with the PMC count of executed VALU instructions (profiles/*_pmc_sq_*) this tells which part of the kernel the
instructions come from: DESIGN.md 4.1 quotes it."""
import re
import subprocess
import sys
import tempfile


def classify(ins: str) -> str:
    op = ins.split()[0]
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_")):
        return "vmul"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main() -> None:
    src, kernel = sys.argv[1], sys.argv[2]
    min_instrs = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    with tempfile.TemporaryDirectory() as d:
        asm = d + "/k.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                        "--cuda-device-only", "-S", "-o", asm, src], check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    start = text.index("\n" + kernel + ":")
    end = text.index(".size\t" + kernel, start)
    blocks, cur = [], ("entry", [])
    for line in text[start:end].split("\n")[2:]:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        t = line.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append(t)
    blocks.append(cur)
    total = {}
    print(f"{'block':<12}{'instrs':>7}  mix")
    for name, ins in blocks:
        mix = {}
        for i in ins:
            c = classify(i)
            mix[c] = mix.get(c, 0) + 1
            total[c] = total.get(c, 0) + 1
        if len(ins) >= min_instrs:
            print(f"{name:<12}{len(ins):>7}  " + " ".join(f"{k}={v}" for k, v in sorted(mix.items())))
    print("whole kernel (static):", " ".join(f"{k}={v}" for k, v in sorted(total.items())))


if __name__ == "__main__":
    main()
