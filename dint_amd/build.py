"""Build the HIP extension in-tree: dint_amd/libdint.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting .so travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdint.so")
SOURCES = ["engine.hip", "k_locks.hip", "k_log.hip", "k_kv.hip", "k_route.hip", "k_bench.hip", "k_txn.hip", "txn_driver.cc", "fasst_client.cc"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SHIM = os.path.join(HERE, "dint_udp_server")
CLIENT = os.path.join(HERE, "dint_udp_client")  # closed-loop loopback load generator (csrc/udp_loop_client.c)
ROCM_LIB = os.environ.get("ROCM_LIB", "/opt/rocm/lib")


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(SHIM) or not os.path.exists(CLIENT):
        return True
    t = os.path.getmtime(LIB)
    inc = os.path.join(HERE, "..", "include")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(inc, f) for f in os.listdir(inc)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + \
          os.environ.get("DINT_CFLAGS", "").split() + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # the UDP host shim (plain C) links against the library it sits next to
    shim = [os.environ.get("CC", "gcc"), "-std=gnu11", "-O2", "-Wall", "-o", SHIM, os.path.join(CSRC, "udp_shim.c"),
            "-L" + HERE, "-ldint", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + ROCM_LIB, "-lpthread"]
    if verbose:
        print(" ".join(shim))
    subprocess.check_call(shim)
    subprocess.check_call([os.environ.get("CC", "gcc"), "-std=gnu11", "-O2", "-Wall", "-pthread", "-o", CLIENT,
                           os.path.join(CSRC, "udp_loop_client.c")])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
