"""Development: bench.py with the tools' RTX_* environment knobs applied to the python binding's defaults first
(bench.py itself reads none).   RTX_ENDGAME=8 python tools/bench_knobs.py --headline-only --steps 20"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _knobs  # noqa: F401,E402

sys.argv[0] = os.path.join(ROOT, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
