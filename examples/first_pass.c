/* first_pass.c -- the hot path from plain C through the C ABI of libmdbg_hip.so: reads (one sequence per line on stdin, or
 * ">header" lines of single-line FASTA) -> minimizers -> palindrome purge -> k-min-mer table of the first pass.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/first_pass.c -o first_pass -Lmetamdbg_amd -lmdbg_hip \
 *       -Wl,-rpath,$PWD/metamdbg_amd -Wl,-rpath,/opt/rocm/lib
 *   ./first_pass < reads.txt
 *
 * Prints the number of minimizers and of table records and the first few records in the layout of
 * kminmerData_abundance.txt (u64 lo, u64 hi, u32 abundance).  Needs an MI355X: the library has no CPU path. */
#define _POSIX_C_SOURCE 200809L   /* getline */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mdbg_hip.h"

static mdbg_ctx *ctx;

static void check(int rc, const char *what) {
    if (rc != MDBG_OK) {
        fprintf(stderr, "%s: %s (code %d)\n", what, mdbg_last_error(ctx), rc);
        exit(1);
    }
}

int main(void) {
    /* read the sequences */
    size_t cap = 1 << 20, n_bases = 0, n_reads = 0, off_cap = 1024;
    char *bases = malloc(cap);
    uint64_t *offsets = malloc(off_cap * sizeof(uint64_t));
    char *line = NULL;
    size_t line_cap = 0;
    ssize_t len;
    offsets[0] = 0;
    while ((len = getline(&line, &line_cap, stdin)) > 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) len--;
        if (len == 0 || line[0] == '>') continue;
        if (n_bases + (size_t)len > cap) { while (n_bases + (size_t)len > cap) cap *= 2; bases = realloc(bases, cap); }
        if (n_reads + 2 > off_cap) { off_cap *= 2; offsets = realloc(offsets, off_cap * sizeof(uint64_t)); }
        memcpy(bases + n_bases, line, (size_t)len);
        n_bases += (size_t)len;
        offsets[++n_reads] = n_bases;
    }
    free(line);

    check(mdbg_create(0, &ctx), "mdbg_create");

    mdbg_reads *reads = NULL;
    check(mdbg_reads_from_ascii(ctx, bases, NULL, offsets, (uint32_t)n_reads, &reads), "mdbg_reads_from_ascii");

    mdbg_scan_params p;
    memset(&p, 0, sizeof p);
    p.minimizer_size = 15;          /* metaMDBG's defaults for HiFi reads */
    p.density = 0.005f;
    p.hpc = 1;
    p.apply_read_filters = 1;
    mdbg_minimizers *mins = NULL, *purged = NULL;
    check(mdbg_scan(ctx, reads, &p, &mins), "mdbg_scan");
    check(mdbg_purge_palindromes(ctx, mins, 4, 11, &purged), "mdbg_purge_palindromes");

    mdbg_table *table = NULL;
    check(mdbg_kminmer_count_first(ctx, purged, 4, 0, &table), "mdbg_kminmer_count_first");

    uint64_t n_min = 0, n_rec = 0, n_solid = 0;
    uint32_t n_r = 0;
    mdbg_minimizers_info(mins, &n_r, &n_min);
    mdbg_table_info(table, NULL, &n_rec, &n_solid, NULL);
    printf("%u reads, %llu bases, %llu minimizers, %llu k-min-mer records (%llu solid)\n", n_r, (unsigned long long)n_bases,
           (unsigned long long)n_min, (unsigned long long)n_rec, (unsigned long long)n_solid);

    uint8_t *records = malloc(n_rec * 20 + 1);
    check(mdbg_table_to_host(ctx, table, records, NULL), "mdbg_table_to_host");
    for (uint64_t i = 0; i < n_rec && i < 5; i++) {
        uint64_t lo, hi;
        uint32_t ab;
        memcpy(&lo, records + 20 * i, 8); memcpy(&hi, records + 20 * i + 8, 8); memcpy(&ab, records + 20 * i + 16, 4);
        printf("  %016llx%016llx  %u\n", (unsigned long long)hi, (unsigned long long)lo, ab);
    }

    free(records);
    mdbg_table_free(table);
    mdbg_minimizers_free(purged);
    mdbg_minimizers_free(mins);
    mdbg_reads_free(reads);
    mdbg_destroy(ctx);
    free(bases);
    free(offsets);
    return 0;
}
