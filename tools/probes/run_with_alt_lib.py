"""Runs a tool against an alternative build of the kernel library: A3D_ALT_LIB=<file under articulation3d_amd/> run_with_alt_lib.py tools/<tool>.py [args]."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import articulation3d_amd._lib as L
alt = os.environ.get("A3D_ALT_LIB")
if alt:
    L.LIB_PATH = os.path.join(ROOT, "articulation3d_amd", alt)
sys.argv = sys.argv[1:]
runpy.run_path(os.path.join(ROOT, sys.argv[0]), run_name="__main__")
