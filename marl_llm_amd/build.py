"""Build libswarmenv.so (the HIP kernels + C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the kernels reproduce the
reference's IEEE-double operation order (no FMA), see csrc/swarm_env.hip.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRCS = [os.path.join(PKG, "csrc", "swarm_env.hip"), os.path.join(PKG, "csrc", "legacy_shim.hip"),
        os.path.join(PKG, "csrc", "policy_mlp.hip")]
INC = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libswarmenv.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libswarmenv.so)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    newest = max([os.path.getmtime(s) for s in SRCS] + [os.path.getmtime(os.path.join(INC, h)) for h in ("swarm_env.h", "swarm_policy.h")])
    return os.path.getmtime(LIB) < newest


def build_lib(force=False, verbose=False, stamps=False):
    """stamps=True builds the DIAGNOSTIC library libswarmenv_stamps.so (per-phase in-kernel clocks, used by
    tools/phase_profile.py); its timings are never quoted as product numbers."""
    out = LIB if not stamps else os.path.join(LIB_DIR, "libswarmenv_stamps.so")
    if not stamps and not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I" + INC] + (["-DSWARM_STAMPS"] if stamps else []) + SRCS + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
