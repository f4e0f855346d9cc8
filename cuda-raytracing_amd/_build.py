"""Build the two in-tree shared libraries of the package.

  librt_hip.so   HIP kernels (gfx950) + the C-ABI of include/rt_hip.h   (hipcc; RCCL is dlopen'ed at run time, not linked)
  librt_host.so  host C++ API mirror + its C facade include/rt_host.h   (g++, links librt_hip.so)

Both are compiled with -ffp-contract=off: results must be bit-identical to the reference's
non-contracted fp32 arithmetic.  The kernels also get -fno-slp-vectorize: hipcc's SLP pass
turns pairs of scalar fp32 ops into v_pk_* instructions, which are slower here (measured +6 % time, +14 VGPRs).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_SO = os.path.join(HERE, "librt_hip.so")
HOST_SO = os.path.join(HERE, "librt_host.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

HIP_SRCS = [os.path.join(CSRC, "rt_kernels.hip"), os.path.join(CSRC, "rt_bvh_build.hip"), os.path.join(CSRC, "rt_comm.hip")]
HIP_DEPS = HIP_SRCS + [os.path.join(CSRC, n) for n in ("rt_math.h", "rt_device_types.h", "rt_scene_internal.h")] + \
    [os.path.join(ROOT, "include", "rt_hip.h")]
HOST_SRCS = [os.path.join(CSRC, "host", n) for n in ("rt_host.cpp", "rt_host_capi.cpp", "rt_image_io.cpp")]


HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]


class BuildError(RuntimeError):
    pass


def kernel_code_hash(sources=None, flags=None):
    """What a committed PMC profile describes and what the loader checks: sha256 over EVERY source librt_hip.so is built from
    (HIP_DEPS: the three translation units, the headers they share -- rt_scene_internal.h is the scene layout all three agree on --
    and include/rt_hip.h) and the compiler flags.  tools/summarize_profile.py stores it with every counters entry and bench.py
    refuses to price a roofline fraction with counters taken from other code (`profile_stale`).  (Rounds 4-5 hashed the kernel
    translation unit and two headers only: an edit to the scene layout or the build kernels left a stale library looking verified.)"""
    import hashlib
    h = hashlib.sha256()
    for path in (sources if sources is not None else HIP_DEPS):
        h.update(os.path.basename(path).encode() + b"\0")
        try:
            h.update(open(path, "rb").read())
        except OSError as e:
            raise BuildError("cannot read %s to hash the library's sources: %s" % (path, e))
    h.update(" ".join(flags if flags is not None else HIP_FLAGS + os.environ.get("RT_HIPCC_EXTRA", "").split()).encode())
    return h.hexdigest()[:16]


HASH_MARKER = b"RT_CODE_HASH="


def library_code_hash(path=None):
    """The kernel code hash compiled into a librt_hip.so (-DRT_CODE_HASH, returned by rt_build_info()), read from the file
    without loading it: the 16 characters after the marker.  None if the file is missing or carries no (or no single) hash."""
    try:
        data = open(path or HIP_SO, "rb").read()
    except OSError:
        return None
    found, at = set(), data.find(HASH_MARKER)
    while at >= 0:
        found.add(data[at + len(HASH_MARKER):at + len(HASH_MARKER) + 16])
        at = data.find(HASH_MARKER, at + 1)
    if len(found) != 1:
        return None
    try:
        return found.pop().decode("ascii")
    except UnicodeDecodeError:
        return None


def variant_allowed():
    """RT_ALLOW_VARIANT_LIB=1 (tools/ab_variants.sh): a librt_hip.so built from other sources or flags than the tree's is used
    as it is; bench.py then prices nothing with the tree's PMC profile, because the line carries the LIBRARY's hash."""
    return os.environ.get("RT_ALLOW_VARIANT_LIB", "")[:1] == "1"


def library_mismatch(path=None):
    """None when the library at `path` was built from the kernel sources and flags of this tree, else a sentence saying what
    differs.  File times are not evidence: a variant copied over the shipped library is newer than every source."""
    want, got = kernel_code_hash(), library_code_hash(path)
    if got == want:
        return None
    return "%s carries kernel code hash %s, the sources in %s hash to %s" % (path or HIP_SO, got or "none", CSRC, want)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _host_deps():
    d = list(HOST_SRCS) + [os.path.join(CSRC, "rt_math.h"), os.path.join(ROOT, "include", "rt_hip.h"),
                           os.path.join(ROOT, "include", "rt_host.h"), HIP_SO]
    hd = os.path.join(CSRC, "host")
    d += [os.path.join(hd, n) for n in os.listdir(hd) if n.endswith((".h", ".hpp"))]
    return d


def build(force=False, verbose=False):
    """Compile whatever is out of date.  Raises CalledProcessError on a compiler error.
    Safe to call from several processes at once (one rank per GPU): an exclusive file lock serialises them
    and every process after the first finds the libraries fresh."""
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    if force or _stale(HIP_SO, HIP_DEPS) or (library_mismatch() is not None and not variant_allowed()):
        # RT_HIPCC_EXTRA: extra compiler flags for A/B experiments on the kernels (e.g. -DRT_PRED_STACK=1); never set in a normal build.
        # The hash of what is being compiled goes into the binary (rt_build_info): the roofline guard of bench.py and libs()
        # check the library that runs, not the sources next to it.
        if not os.path.exists(HIPCC):
            raise RuntimeError("librt_hip.so is missing or was not built from this tree (%s) and %s does not exist to rebuild it"
                               % (library_mismatch() or "stale", HIPCC))
        cmd = [HIPCC] + HIP_FLAGS + os.environ.get("RT_HIPCC_EXTRA", "").split() + ['-DRT_CODE_HASH="%s"' % kernel_code_hash()] + \
              ["-o", HIP_SO] + HIP_SRCS
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    if force or _stale(HOST_SO, _host_deps()):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-D__HIP_PLATFORM_AMD__",
               "-I" + os.path.join(ROCM, "include"), "-o", HOST_SO] + HOST_SRCS + \
              ["-L" + HERE, "-lrt_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return HIP_SO, HOST_SO


if __name__ == "__main__":
    if "--print-hash" in sys.argv:                               # (CMakeLists.txt, at BUILD time: the definition it passes to the kernels,
        flags = None                                             #  hashed with the flags CMake itself compiles with: --flags "...")
        if "--flags" in sys.argv:
            flags = sys.argv[sys.argv.index("--flags") + 1].split()
        text = kernel_code_hash(flags=flags)
        if "--header" in sys.argv:                               # --header <file>: written only when the hash changed (no needless rebuilds)
            path = sys.argv[sys.argv.index("--header") + 1]
            line = '#define RT_CODE_HASH "%s"\n' % text
            if not os.path.exists(path) or open(path).read() != line:
                open(path, "w").write(line)
        else:
            print(text)
        sys.exit(0)
    build(force="--force" in sys.argv, verbose=True)
