"""Run a script against another build of the library: python scripts/with_lib.py <libpath> <script> [args...]
(tuning aid for compile-time variants, e.g. a library built with EXTRA=-DCOMMS_OS1024_SWAP=1)."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import comms_rs_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
