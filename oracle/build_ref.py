"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (vilco_amd/).

Recipe that compiles the reference's own 1-D NMS extension *from the source where it
lies* (/root/reference/MQ/libs/utils/csrc/nms_cpu.cpp, one translation unit, ATen +
pybind11 only) into oracle/_ref/nms_1d_cpu.so.  Nothing is copied into the repo; the
output directory is git-ignored but travels to the GPU box with the snapshot, where it
serves as the "reference" checker/baseline for the HIP NMS (tests/, bench.py cpu_baseline).

We do not run the reference's setup.py: the two flags it passes (CppExtension defaults
+ -fopenmp, MQ/libs/utils/setup.py:8-18) are restated here.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/MQ/libs/utils/csrc/nms_cpu.cpp"
OUT_DIR = os.path.join(HERE, "_ref")


def ref_so_path():
    return os.path.join(OUT_DIR, "nms_1d_cpu.so")


def build(verbose=False):
    """Build oracle/_ref/nms_1d_cpu.so if the reference source is present.

    Returns the path of the .so, or None when /root/reference is absent (GPU box) and
    no prebuilt file exists."""
    so = ref_so_path()
    if not os.path.exists(REF_SRC):
        return so if os.path.exists(so) else None
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(REF_SRC):
        return so
    os.makedirs(OUT_DIR, exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name="nms_1d_cpu", sources=[REF_SRC], build_directory=OUT_DIR,
         extra_cflags=["-O2", "-fopenmp"], verbose=verbose)
    return so if os.path.exists(so) else None


def load_ref():
    """Import the prebuilt reference module (python module `nms_1d_cpu`), or None."""
    so = build()
    if so is None:
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("nms_1d_cpu", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(verbose=True)
    print("reference nms extension:", p)
    sys.exit(0 if p else 1)
