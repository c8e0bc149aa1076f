"""Build libpcr_hip.so (hand-written gfx950 kernels + the C ABI of include/pcr.h) IN-TREE with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpcr_hip.so")
OBJ = os.path.join(HERE, "build")

SOURCES = ["api.cpp", "numerics.cpp", "icp.cpp", "comm.cpp", "nn1_brute.hip", "kabsch.hip", "plane.hip",
           "search_f64.hip", "grid.hip", "voxel.hip", "iss.hip", "desc.hip", "ground.hip", "knn_grid.hip", "radius_grid.hip", "p2plane.hip", "sort.hip"]

# -ffp-contract=off: the distance arithmetic contract is UNFUSED (nanoflann.hpp:403-406 / kdtree.hpp:341-346);
#   one fma changes d2 in the last bit and flips near-tie winners.
# -fno-slp-vectorize: keeps the inner loop on plain v_sub/v_mul/v_add_f32; the SLP vectoriser otherwise packs
#   pairs into v_pk_mul_f32 / v_pk_add_f32, which issue at half rate on gfx950 and need extra v_mov to form
#   register pairs (measured: profiles/).
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950's register file is unified) — the brute-force filter takes the minimum
# of its 16 accumulators with the vector ALU, and the AGPR form would cost one v_accvgpr_read per accumulator (nn1_brute.hip, MTRACK)
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(a: str, b: str) -> bool:
    return os.path.exists(a) and os.path.getmtime(a) >= os.path.getmtime(b)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    # every object depends on every internal header (a handful of small files: precise tracking is not worth a stale object)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")) + [os.path.join(ROOT, "include", "pcr.h"), __file__]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        if force or not _newer(obj, src) or any(not _newer(obj, h) for h in headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [cc, *FLAGS, *os.environ.get("PCR_EXTRA_FLAGS", "").split(), "-c", src, "-o", obj]      # A/B builds: PCR_EXTRA_FLAGS="-DPCR_..."
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


def build_host_sanitized(out_dir: str) -> str:
    """ASan + UBSan on the HOST side only (-Xarch_host; GPU sanitizers are not available on the pool): a second library
    for running the host-logic tests on the CPU:
        PCR_LIB_PATH=<out>/libpcr_hip_san.so ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
        LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) \
        python -m pytest tests -m "not gpu" --deselect tests/test_abi.py::test_header_is_plain_c_and_links_from_c"""
    os.makedirs(out_dir, exist_ok=True)
    cc = hipcc()
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=undefined"]
    flags = [f for f in FLAGS if f != "-O3"] + ["-O1", "-g"] + san
    objs = []
    for s in SOURCES:
        obj = os.path.join(out_dir, s + ".o")
        r = subprocess.run([cc, *flags, "-c", os.path.join(CSRC, s), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        objs.append(obj)
    out = os.path.join(out_dir, "libpcr_hip_san.so")
    r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", out, *objs, "-ldl"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return out


if __name__ == "__main__":
    if "--sanitize-host" in sys.argv:
        print(build_host_sanitized(sys.argv[sys.argv.index("--sanitize-host") + 1]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
