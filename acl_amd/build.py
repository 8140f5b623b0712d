"""Builds every native piece in-tree (no JIT cache): the HIP library for gfx950, the host-only synthetic
clip writer, and the test-only CPU oracle. hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "acl_amd", "csrc")
LIB = os.path.join(ROOT, "acl_amd", "lib")
ORACLE = os.path.join(ROOT, "oracle")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the decode must not fuse multiply-add (parity with the reference's unfused SSE arithmetic)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wextra", "-ldl",
             # kernel arguments arrive in SGPRs instead of behind a first scalar load (gfx940+)
             "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    proc = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build step failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    return proc.stdout


def build_synth(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libaclsynth.so")
    sources = [os.path.join(CSRC, f) for f in ("clip_synth.cpp", "clip_synth.h", "acl_format.h")]
    if force or _newer(target, sources):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", sources[0], "-o", target])
    return target


def _hip_sources():
    # one translation unit: aclhip.hip includes its parts (*.inl)
    sources = [os.path.join(CSRC, f) for f in ("aclhip.hip", "aclhip_device.h", "acl_format.h")] + [os.path.join(ROOT, "include", "aclhip.h")]
    return sources + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inl"))


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libaclhip.so")
    sources = _hip_sources()
    if force or _newer(target, sources):
        _run([HIPCC] + HIP_FLAGS + [sources[0], "-o", target])
    return target


def build_hip_lab(force=False):
    """libaclhip_lab.so: the same library with -DACLHIP_LAB_KNOBS -- it reads the measurement knobs and ACLHIP_SHORT_EXACT_MATH=1
    (host_context.inl: lab_knob), which the shipped library ignores. Tests that need such a knob and tools/ point ACLHIP_LIBRARY at it."""
    os.makedirs(LIB, exist_ok=True)
    target = os.path.join(LIB, "libaclhip_lab.so")
    sources = _hip_sources()
    if force or _newer(target, sources):
        _run([HIPCC] + HIP_FLAGS + ["-DACLHIP_LAB_KNOBS", sources[0], "-o", target])
    return target


def build_oracle(force=False):
    """The CPU oracle (test infrastructure) and, when /root/reference is present, oracle/_ref (the reference's own headers)."""
    if force:
        _run(["make", "-C", ORACLE, "clean"])
    _run(["make", "-C", ORACLE, "all"])
    return os.path.join(ORACLE, "libacloracle.so")


def build_all(force=False):
    # the two builds of the HIP library (half a minute each) side by side
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=2) as pool:
        lab = pool.submit(build_hip_lab, force)
        hip = build_hip(force)
        lab = lab.result()
    return {"synth": build_synth(force), "hip": hip, "hip_lab": lab, "oracle": build_oracle(force)}


if __name__ == "__main__":
    for name, path in build_all().items():
        print(name, path)
