"""Build libphc_hip.so (gfx950 only) in-tree with hipcc.

    python -m phc_gnn_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot.  There is no JIT
and no fallback: if the library is absent the Python layer refuses to run.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO, "include")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "libphc_hip.so")
ARCH = "gfx950"

SOURCES = ["api.hip", "phm_linear.hip", "gemm_agw.hip", "aggregate.hip", "csr.hip", "encoder.hip", "norm.hip", "softmax_agg.hip", "readout.hip", "optim.hip", "downstream.hip", "head_chain.hip", "mlp_fused.hip"]
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{CSRC}",
                "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value", "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libphc_hip.so cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, defines=(), lib_path: str = LIB_PATH, obj_dir: str = OBJ_DIR) -> str:
    """``defines`` / ``lib_path`` / ``obj_dir``: an experiment variant next to the product library (A/B runs on one box
    through PHC_LIB_PATH, e.g. ``python -m phc_gnn_amd.build --variant noxcd -DPHC_XCD_MAP=0``)."""
    hipcc = _hipcc()
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "phc_hip.h"))
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def compile_one(src: str) -> str:
        path = os.path.join(CSRC, src)
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc, "-c", "-x", "hip", path, "-o", obj] + COMMON_FLAGS + list(defines)
            if verbose:
                print("[phc build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    # The H-stationary GEMM's inline-asm contract (gemm_hs_device.h) is audited in the generated code whenever its translation
    # unit is recompiled (scripts/audit_hs_asm_loads.py: prologue loads untouched until their counted wait, DMA instruction
    # counts and counted waits of the ring as the source derives them); a violation FAILS the build.  Runs beside the compiles.
    hs_obj = os.path.join(obj_dir, "phm_linear.o")
    hs_deps = [os.path.join(CSRC, "phm_linear.hip")] + headers
    stamp = os.path.join(obj_dir, "hs_asm_audit.ok")
    need_audit = not defines and (force or _stale(hs_obj, hs_deps) or _stale(stamp, hs_deps)) and os.environ.get("PHC_SKIP_ASM_AUDIT") != "1"

    def audit_hs() -> str:
        import importlib.util
        import tempfile
        spec = importlib.util.spec_from_file_location("audit_hs_asm_loads", os.path.join(REPO, "scripts", "audit_hs_asm_loads.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        with tempfile.TemporaryDirectory() as d:
            asm = os.path.join(d, "phm_linear.s")
            mod.compile_asm(asm)
            text = open(asm).read()
        loads, ring = mod.audit(text), mod.audit_ring(text)
        bad = [f"{name}: {b}" for name, _, _, bl in loads for b in bl] + [f"{name}: {pr}" for name, prs in ring for pr in prs]
        if bad or not loads or not ring:
            raise RuntimeError("H-stationary GEMM asm audit failed (hipcc " + _hipcc() + "):\n  " + "\n  ".join(bad[:12] or ["no variants found"]))
        return f"{len(loads)} variants with asm loads, {len(ring)} instantiations: ok"

    with ThreadPoolExecutor(max_workers=4) as ex:
        audit_future = ex.submit(audit_hs) if need_audit else None
        objs = list(ex.map(compile_one, sources))
        if audit_future is not None:
            msg = audit_future.result()
            open(stamp, "w").write(msg + "\n")
            if verbose:
                print("[phc build] asm audit:", msg, flush=True)
    if force or _stale(lib_path, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib_path] + objs
        if verbose:
            print("[phc build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        out_dir = os.path.join(REPO, "scripts", "exp", "variants")
        os.makedirs(out_dir, exist_ok=True)
        print(build(force=True, defines=[a for a in sys.argv if a.startswith("-D")],
                    lib_path=os.path.join(out_dir, f"libphc_hip_{name}.so"), obj_dir=os.path.join(OBJ_DIR, "variant_" + name)))
    else:
        print(build(force="--force" in sys.argv))
