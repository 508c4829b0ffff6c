import sys, os
import os as _o, sys as _s; _s.path.insert(0, _o.path.dirname(_o.path.dirname(_o.path.dirname(_o.path.abspath(__file__)))))
sys.argv = ["bench.py", "--no-secondary", "--no-cpu-baseline"]
mode = os.environ.get("EXP", "")
import crypto_amd as ca
from crypto_amd import _native
if mode == "twin_init_first":
    ca.init(0)
    with ca.twin():
        pass
elif mode == "twin_load_only":
    _native.dev_lib()
import runpy
runpy.run_path(_o.path.join(_s.path[0], "bench.py"), run_name="__main__")
