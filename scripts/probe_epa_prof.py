"""Runs config 4 on the TOHC_EPA_PROF build of the library (scripts/native/libtohc_prof.so) and lets it print its
in-kernel EPA cycle profile (diagnostic only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajopt_amd import engine
engine.LIB_PATH = os.path.join(ROOT, "scripts", "native", "libtohc_prof.so")
import runpy
sys.argv = [sys.argv[0]] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "scripts", "probe_latency.py"), run_name="__main__")
