"""bench.py on the -DVT_ABLATIONS build of the library (libvitron_hip_abl.so), so that the environment switches of the A/B variants
(VT_W4_EPI_DIRECT, VT_FLASH_ABL, ...) apply to the whole benchmark step: python tools/bench_abl.py <bench.py arguments>.
Measurement helper; the product benchmark is bench.py on libvitron_hip.so."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vitron_amd import _lib  # noqa: E402

_lib.load(ablations=True)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
