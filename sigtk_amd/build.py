"""Build libsigtk_gpu.so (HIP kernels + C ABI) and the sigtk-amd host CLI, in-tree.

hipcc cross-compiles for gfx950 without a GPU present.  Numerics-critical flags:
  -ffp-contract=off   the reference is plain C99 on SSE2 (one rounding per operator); HIP's
                      default 'fast' contraction would fuse mul+add into FMA and change bits
  (f32 divide/sqrt stay correctly rounded: -fhip-fp32-correctly-rounded-divide-sqrt is the
   hipcc default and is passed explicitly)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB = os.path.join(PKG, "libsigtk_gpu.so")
CLI = os.path.join(PKG, "sigtk-amd")
CLI_ASAN = os.path.join(PKG, "sigtk-amd-asan")

HIP_SOURCES = ["api.hip", "api_stat.hip", "event_kernels.hip", "stat_kernels.hip", "misc_kernels.hip",
               "svb_kernels.hip", "inflate_kernels.hip", "ent_kernels.hip", "qts_kernels.hip", "job.hip", "shims.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-Wno-c++20-extensions", "-Wno-pass-failed"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """One object per .hip source (in parallel, rebuilt only when the source or a header changed), then one link.
    Without -fgpu-rdc every translation unit's device code is self-contained, exactly as when hipcc is handed all the
    sources at once -- only faster to iterate on (event_kernels.hip and stat_kernels.hip take ~50 s each)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "sigtk_gpu.h"))
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    jobs = []
    objs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            jobs.append([cc, *flags, "-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(run, jobs))
    if force or jobs or not _newer(LIB, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


def build_variant(tag: str, defines, sources=("event_kernels.hip", "api.hip"), verbose: bool = False) -> str:
    """A development variant of the library: `sources` recompiled with the given -D flags, everything else taken from
    the shipped build's objects -> sigtk_amd/_variants/libsigtk_gpu_<tag>.so (select it with SIGTK_AMD_LIB=<path>).
    -DSGK_DEV=1 is the instrumented build of event_args.h (phases switched off per call, per-wave timestamps)."""
    build_lib(verbose=verbose)
    vdir = os.path.join(PKG, "_variants")
    os.makedirs(vdir, exist_ok=True)
    cc = hipcc()
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(defines)
    objs = []
    for s_ in HIP_SOURCES:
        base = s_[:-4]
        if s_ in sources:
            obj = os.path.join(vdir, "%s_%s.o" % (base, tag))
            cmd = [cc, *flags, "-c", "-o", obj, os.path.join(CSRC, s_)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        else:
            obj = os.path.join(PKG, "build", base + ".o")
        objs.append(obj)
    out = os.path.join(vdir, "libsigtk_gpu_%s.so" % tag)
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
    return out


def build_cli(force: bool = False, verbose: bool = False) -> str:
    if not os.path.isdir(HOST):
        return ""
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".c"))
    if not srcs:
        return ""
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "sigtk_gpu.h"))
    if not force and _newer(CLI, deps) and _newer(CLI, [LIB]):
        return CLI
    cmd = ["gcc", "-O2", "-std=c99", "-D_GNU_SOURCE", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", CLI, *srcs,
           "-L", PKG, "-lsigtk_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lm", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI


def build_cli_asan(force: bool = False, verbose: bool = False) -> str:
    """The host sources (CLI, BLOW5 reader, formatter) under -fsanitize=address,undefined -- the counterpart of the
    reference's `make asan=1` (Makefile:31-34).  CPU side only: the GPU library is linked as it is (GPU
    AddressSanitizer is not available on this pool); tests/test_cli_cpu.py runs the reader's hostile-input cases
    through this binary."""
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".c"))
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "sigtk_gpu.h"))
    if not os.path.exists(LIB):
        build_lib(verbose=verbose)
    if not force and _newer(CLI_ASAN, deps) and _newer(CLI_ASAN, [LIB]):
        return CLI_ASAN
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-std=c99", "-D_GNU_SOURCE", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", CLI_ASAN, *srcs,
           "-L", PKG, "-lsigtk_gpu", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lm", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return CLI_ASAN


def build_tools(force: bool = False, verbose: bool = False) -> None:
    """the stand-alone HIP programs of tools/ (issue-rate microbenchmark, hardware accuracy check of v_rsq_f32)"""
    tdir = os.path.join(ROOT, "tools")
    for name in ("rsq_check", "valu_rate", "fetch_calib"):
        src = os.path.join(tdir, name + ".hip")
        exe = os.path.join(tdir, name)
        deps = [src] + ([os.path.join(tdir, "valu_rate_tests.inc")] if name == "valu_rate" else [])
        if not os.path.exists(src) or (not force and _newer(exe, deps)):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O2", "-o", exe, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m sigtk_amd.build --variant <tag> -DNAME=VALUE ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [x for x in sys.argv[i + 2:] if x.startswith("-D")], verbose=True))
        sys.exit(0)
    build_tools(force="--force" in sys.argv, verbose=True)
    build_lib(force="--force" in sys.argv, verbose=True)
    build_cli(force="--force" in sys.argv, verbose=True)
    if "--asan" in sys.argv:
        build_cli_asan(force="--force" in sys.argv, verbose=True)
