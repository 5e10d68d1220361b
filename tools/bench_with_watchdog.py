"""python tools/bench_with_watchdog.py SECONDS [bench.py flags]: bench.py with every thread's Python stack dumped to stderr
after SECONDS (then exit) -- where a multi-threaded run sits when it does not come back."""
import faulthandler
import os
import runpy
import sys

secs = float(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
