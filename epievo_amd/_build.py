"""In-tree builds of the native pieces (no JIT cache, no pip install).

  build_hip()    hipcc --offload-arch=gfx950  -> epievo_amd/libepievo_mi355x.so
                 (the C-ABI of include/epievo_mi355x.h: HIP kernels + host glue)
  build_comm()   hipcc (host code) + librccl  -> epievo_amd/libepv_rccl.so
                 (the C-ABI of include/epievo_mi355x_comm.h: halo exchange + statistics all-gather)
  build_host()   g++                          -> epievo_amd/libepv_host.so
                 (model / M-step / file formats / synthetic-input simulator)
  build_driver() g++                          -> epievo_amd/libepv_driver.so
                 (the C-ABI of include/epievo_mi355x_driver.h: epv::SingleSiteSampler, the C++ EM
                 driver of the CLIs, for bench.py and the tests; links the two libraries above)
  build_cli()    g++                          -> epievo_amd/bin/epievo_*  (drop-in CLIs)
  build_oracle() make -C oracle [ref]         -> oracle/liborc.so (+ oracle/_ref/…)
                 TEST INFRASTRUCTURE ONLY; the product never loads it.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(CSRC, "host")
INCLUDE = os.path.join(ROOT, "include")

HIP_SO = os.path.join(PKG, "libepievo_mi355x.so")
COMM_SO = os.path.join(PKG, "libepv_rccl.so")
HOST_SO = os.path.join(PKG, "libepv_host.so")
DRIVER_SO = os.path.join(PKG, "libepv_driver.so")
BIN_DIR = os.path.join(PKG, "bin")

HOST_SOURCES = ["epv_model.cpp", "epv_sim.cpp", "epv_io.cpp", "epv_indep.cpp", "epv_forward.cpp", "epv_host_abi.cpp"]
# the reference library is built -O3 without -march (no FMA contraction); match it
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-pthread"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def hipcc_path():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force=False):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))
            if f.endswith((".hip", ".h", ".hpp"))]
    srcs.append(os.path.join(INCLUDE, "epievo_mi355x.h"))
    if not force and _newer(HIP_SO, srcs):
        return HIP_SO
    units = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # bit-parity with the CPU oracle: no implicit FMA contraction, IEEE div/sqrt
           "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
           "-I", INCLUDE, "-I", CSRC, "-o", HIP_SO] + units
    _run(cmd)
    return HIP_SO


def build_comm(force=False):
    """the exchange layer (include/epievo_mi355x_comm.h): host code on the HIP runtime, RCCL
    linked directly.  Its own library so that the kernels' library has no RCCL dependency."""
    src = os.path.join(CSRC, "comm", "epv_comm.cpp")
    deps = [src, os.path.join(INCLUDE, "epievo_mi355x_comm.h"), os.path.join(INCLUDE, "epievo_mi355x.h")]
    if not force and _newer(COMM_SO, deps):
        return COMM_SO
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path()))), "lib")
    _run([hipcc_path(), "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
          "-I", INCLUDE, "-o", COMM_SO, src, "-L", rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib])
    return COMM_SO


def build_host(force=False):
    srcs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".cpp", ".hpp"))]
    if not force and _newer(HOST_SO, srcs):
        return HOST_SO
    cmd = ["g++"] + HOST_FLAGS + ["-shared", "-I", HOST, "-I", INCLUDE, "-o", HOST_SO] + \
        [os.path.join(HOST, f) for f in HOST_SOURCES]
    _run(cmd)
    return HOST_SO


def build_driver(force=False):
    srcs = [os.path.join(HOST, f) for f in ("epv_driver_abi.cpp", "epv_sampler.cpp")]
    deps = srcs + [os.path.join(HOST, "epv_sampler.hpp"), os.path.join(INCLUDE, "epievo_mi355x_driver.h"), HIP_SO, COMM_SO]
    if not force and _newer(DRIVER_SO, deps):
        return DRIVER_SO
    _run(["g++"] + HOST_FLAGS + ["-shared", "-I", HOST, "-I", INCLUDE, "-o", DRIVER_SO] + srcs +
         ["-L", PKG, "-lepievo_mi355x", "-lepv_rccl", "-Wl,-rpath,$ORIGIN"])
    return DRIVER_SO


def build_oracle(with_ref=True):
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-C", odir])
    if with_ref and os.path.isdir("/root/reference/src/libepievo"):
        _run(["make", "-C", odir, "-j8", "ref"])
    return os.path.join(odir, "liborc.so")


def build_all(force=False):
    build_host(force)
    build_hip(force)
    build_comm(force)
    build_driver(force)
    cli = os.path.join(HOST, "cli")
    if os.path.isdir(cli):
        build_cli(force)
    build_oracle()


def build_cli(force=False):
    cli = os.path.join(HOST, "cli")
    os.makedirs(BIN_DIR, exist_ok=True)
    common = [os.path.join(HOST, f) for f in ("epv_model.cpp", "epv_sim.cpp", "epv_io.cpp", "epv_indep.cpp",
                                              "epv_forward.cpp", "epv_sampler.cpp", "epv_options.cpp")]
    outs = []
    for f in sorted(os.listdir(cli)):
        if not f.endswith(".cpp"):
            continue
        out = os.path.join(BIN_DIR, f[:-4])
        srcs = [os.path.join(cli, f)] + common
        if force or not _newer(out, srcs + [HIP_SO, COMM_SO, os.path.join(HOST, "epv_sampler.hpp")]):
            _run(["g++"] + [x for x in HOST_FLAGS if x not in ("-fPIC", "-fvisibility=hidden")] +
                 ["-I", HOST, "-I", INCLUDE, "-o", out] + srcs +
                 ["-pthread", "-L", PKG, "-lepievo_mi355x", "-lepv_rccl", "-Wl,-rpath,$ORIGIN/.."])
        outs.append(out)
    return outs
