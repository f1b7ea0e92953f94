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
SOURCES = ["engine.hip", "k_locks.hip", "k_log.hip", "k_kv.hip", "k_kv_tatp.hip", "k_kv_store.hip", "k_kv_smallbank.hip", "k_route.hip", "k_bench.hip", "k_txn.hip", "txn_driver.cc", "fasst_client.cc"]
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


def _compile(job):
    src, obj, flags, verbose, fresh = job
    if fresh:
        return obj
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    # one object per source, compiled side by side (k_kv.hip alone is ~2 minutes: the others hide behind it), then one link
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = os.environ.get("DINT_CFLAGS", "").split()
    # an object is reused while it is newer than its source and every header (force: everything again -- what build() of
    # __graft_entry__ asks for, and what a change of DINT_CFLAGS needs)
    inc = os.path.join(HERE, "..", "include")
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(inc, f) for f in os.listdir(inc)]
    t_h = max(os.path.getmtime(h) for h in hdrs)

    def fresh(src, obj):
        return (not force and not flags and os.path.exists(obj) and os.path.getmtime(obj) > max(t_h, os.path.getmtime(src)))

    jobs = [(os.path.join(CSRC, s), os.path.join(objdir, os.path.splitext(s)[0] + ".o"), flags, verbose) for s in SOURCES]
    jobs = [j + (fresh(j[0], j[1]),) for j in jobs]
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
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
    import sys

    print(build(force="--force" in sys.argv, verbose=True) if "--force" in sys.argv or _stale() else LIB)
