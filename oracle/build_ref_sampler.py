"""Recipe for oracle/_ref/pv2_ref_smooth_sampler.so: the REFERENCE's own trilinear sampler
(libs/smooth-sampler/smooth_sampler/csrc/{smooth_sampler.cpp, smooth_sampler_kernel.cu}) compiled for
gfx950 through torch's hipify, from the sources where they lie under /root/reference.

TEST INFRASTRUCTURE ONLY (tests/test_gpu_sampler_vs_reference_binary.py): the product sampler is
ponderv2_amd/csrc/trilinear.hip, written from scratch; this binary is the reference's arithmetic to
compare it with on the MI355X.  Nothing of the reference is copied into the repository: the two source
files are staged in a temporary directory outside the repo (hipify writes next to its inputs, and
/root/reference is read-only), only the built shared object lands in oracle/_ref/ (git-ignored, travels
with gpurun snapshots).  GPU-only (smooth_sampler.cpp:8-10 rejects host tensors), so it cannot serve as a
CPU baseline.  Usage: python oracle/build_ref_sampler.py [reference root]
"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "pv2_ref_smooth_sampler.so")
NAME = "pv2_ref_smooth_sampler"


def build(ref_root="/root/reference", force=False):
    src_dir = os.path.join(ref_root, "libs", "smooth-sampler", "smooth_sampler", "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("smooth_sampler.cpp", "smooth_sampler_kernel.cu")]
    if not all(os.path.exists(s) for s in srcs):
        return None          # no reference checkout here (the GPU box): use the prebuilt file, if any
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= max(os.path.getmtime(s) for s in srcs + [__file__])):
        return OUT
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils import cpp_extension

    work = tempfile.mkdtemp(prefix="pv2_ref_sampler_")
    try:
        staged = [shutil.copy(s, work) for s in srcs]
        bdir = os.path.join(work, "build")
        os.makedirs(bdir)
        cpp_extension.load(name=NAME, sources=staged, build_directory=bdir, is_python_module=False,
                           extra_cflags=["-O2"], extra_cuda_cflags=["-O2"], verbose=False)
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        shutil.copy(os.path.join(bdir, NAME + ".so"), OUT)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return OUT


def load():
    """The built module (torch must see a GPU to call into it), or None when it was never built."""
    if not os.path.exists(OUT):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)

    spec = importlib.util.spec_from_file_location(NAME, OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(*(sys.argv[1:2] or ["/root/reference"]), force="--force" in sys.argv))
