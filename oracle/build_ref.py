#!/usr/bin/env python3
"""Build the REAL reference kernel module into oracle/_ref/ (test infrastructure only).

The reference's only native code is ``velocyto/speedboosted.pyx`` (Cython + OpenMP,
built by the reference's ``setup.py:17-21`` with ``-fopenmp -ffast-math``).  This recipe
cythonizes that file *from where it lies* under /root/reference (no source is copied
into the repo; the generated .c and the .so go to oracle/_ref/, which is git-ignored)
and compiles it with the reference's own flags.  The resulting extension module

    oracle/_ref/speedboosted.cpython-310-x86_64-linux-gnu.so

is a build PRODUCT (oracle/_ref/ is listed in .gitignore: never in history; it is NOT gpurun-ignored, so it travels to the GPU box
with the repository snapshot like the library's own .so) and is used, always in a subprocess (oracle.reference_coldeltacor), only
  * to validate the C restatement in oracle/velocyto_oracle.c and to generate tests/golden/*.npz (tests/golden/make_golden.py);
  * by tests/ where it is present: the restatement and the HIP kernels against the reference's own kernels on fresh random inputs;
  * by bench.py's ``cpu_baseline`` leg: stage D - 98 % of the CPU time of the path - timed with the reference's own kernel on the
    GPU box's host cores (cpu_baseline.kind = "reference"); where the module is absent the pinned restatement is timed (kind "port").

Runs only where /root/reference exists (this container).  Needs Cython (3.2.9 here),
gcc and numpy headers - all present in the image; nothing is stubbed.
"""
import os
import subprocess
import sys
import sysconfig

REF_PYX = "/root/reference/velocyto/speedboosted.pyx"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def build(force: bool = False) -> str:
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    so = os.path.join(OUT, "speedboosted" + ext)
    def mark():
        # the opt-in the tests and smoke() read (oracle.reference_module_expected): this tree was given the reference kernels, for this ABI
        import json
        with open(os.path.join(OUT, "built.json"), "w") as f:
            json.dump({"ext_suffix": ext, "module": os.path.basename(so)}, f)
        return so

    if os.path.exists(so) and not force:
        return mark()
    if not os.path.exists(REF_PYX):
        raise FileNotFoundError(f"{REF_PYX} not present: the reference only exists in the build container")
    c_file = os.path.join(OUT, "speedboosted.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", REF_PYX, "-o", c_file])
    inc = sysconfig.get_paths()["include"]
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffast-math", "-fwrapv", "-w",
           "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
           f"-I{inc}", f"-I{np.get_include()}", c_file, "-o", so]
    subprocess.check_call(cmd)
    os.remove(c_file)  # generated from reference source: keep only the binary
    return mark()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
