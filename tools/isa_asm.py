"""gfx950 assembly of a csrc/*.hip file (hipcc -save-temps with the library's own flags), cached per process tree: the ISA audits
(check_dma_barriers.py, check_mfma_war.py; run by tests/test_host_api.py) read the same files -- compile each once, all of them in parallel."""
import concurrent.futures as cf
import glob
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_CACHE = os.path.join(tempfile.gettempdir(), "xfh_isa_cache")


def _key(src, flags):
    h = hashlib.sha256()
    h.update(open(src, "rb").read())
    for dep in sorted(glob.glob(os.path.join(os.path.dirname(src), "*.hpp"))) + [os.path.join(ROOT, "include", "xfeat_hip.h")]:
        h.update(open(dep, "rb").read())
    h.update(" ".join(flags).encode())
    return os.path.join(_CACHE, os.path.basename(src) + "." + h.hexdigest()[:16] + ".s")


def _flags(src):
    from accelerated_features_amd.build import EXTRA_FLAGS
    return ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on"] + EXTRA_FLAGS.get(os.path.basename(src), [])


def asm(src):
    flags = _flags(src)
    out = _key(src, flags)
    if not os.path.exists(out):
        os.makedirs(_CACHE, exist_ok=True)
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-save-temps", "-c", src, "-o", os.path.join(td, "o.o")], cwd=td, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(r.stderr)
            text = open(glob.glob(os.path.join(td, "*gfx950.s"))[0]).read()
        tmp = out + f".{os.getpid()}.tmp"
        open(tmp, "w").write(text)
        os.replace(tmp, out)
    return open(out).read()


def prefetch(files):
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(files)))) as ex:
        list(ex.map(asm, files))
