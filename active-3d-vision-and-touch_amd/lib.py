"""ctypes binding of ``liba3vt.so`` (the C ABI declared in ``include/a3vt.h``).

There is deliberately NO fallback: if the shared library is missing or a call fails, a
``RuntimeError`` is raised.  ``build()`` compiles the HIP sources in-tree with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("A3VT_LIB", os.path.join(_HERE, "liba3vt.so"))  # A3VT_LIB: developer override (variant builds)
SOURCES = ["capi.hip", "gcn_gemm.hip", "gcn_gemmw.hip", "gcn_dww.hip", "gcn_gemm16.hip", "gcn_gemm3.hip", "gcn_csr.hip", "gcn_csrq.hip", "gcn_csrqs.hip", "gcn_bf16s.hip", "posenc.hip", "posenc_wide.hip", "bias_grad.hip", "bnrelu.hip", "conv5.hip", "adam.hip", "sample.hip", "chamfer.hip", "nn_prune.hip",
           "pooling.hip"]
# Per-file extra flags (none at present; sample.hip / gcn_csr.hip rely on IEEE NaN semantics — the reference's NaN
# scrubs, a3vt_check_finite — so fast-math style flags must never be applied globally).
# chamfer.hip: keep the nearest-neighbour loop on scalar fp32 ops (the SLP vectoriser would re-pack it into v_pk_*_f32,
# which issues at half the rate on gfx950 and needs s_nop hazard padding)
# gcn_csrq.hip: scalar fma chains keep one register per edge weight (v_pk_fma_f32 wants (w, w) pairs): see the file header
EXTRA_FLAGS = {"chamfer.hip": ["-fno-slp-vectorize"], "nn_prune.hip": ["-fno-slp-vectorize"],
               "gcn_csrq.hip": ["-fno-slp-vectorize"], "gcn_csrqs.hip": ["-fno-slp-vectorize"],
               "gcn_gemmw.hip": ["-std=c++20"], "gcn_dww.hip": ["-std=c++20"]}   # (templated lambdas over the chunk / column-tile index)

_vp, _i, _sz, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64


class AdjSplit(ctypes.Structure):
    """``a3vt_adj_split`` (include/a3vt.h): the fused vision + touch adjacency as P + a complete bipartite block."""
    _fields_ = [("rowptr", _vp), ("col", _vp), ("scale", _vp), ("cls", _vp),
                ("max_degree", ctypes.c_int32), ("n_seam", ctypes.c_int32), ("n_centre", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/a3vt.h one to one.
SIGNATURES = {
    "a3vt_version": (_i, []),
    "a3vt_last_error": (ctypes.c_char_p, []),
    "a3vt_csr_validate": (_i, [_vp, _vp, _i, _i]),
    "a3vt_gcn_stack_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "a3vt_gcn_stack_mask_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "a3vt_gcn_stack_scratch_bytes_mode": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "a3vt_gcn_stack_stash_bytes": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp]),
    "a3vt_gcn_stack_fwd": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_gcn_stack_bwd": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_gcn_stack_bwd_acc": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "a3vt_adj_split_validate": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "a3vt_gcn_stack_fwd_adj": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_gcn_stack_bwd_adj": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "a3vt_gcn_layer_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "a3vt_gcn_layer_fwd": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "a3vt_gcn_layer_bwd": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i,
                                _vp, _vp, _vp, _vp, _vp]),
    "a3vt_wt_rows": (_i, [_i]),
    "a3vt_wt_ld": (_i, [_i]),
    "a3vt_transpose_weight": (_i, [_vp, _i, _i, _vp, _vp]),
    "a3vt_rowgemm": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "a3vt_posenc_param_count": (_sz, [_i]),
    "a3vt_posenc_scratch_bytes": (_sz, [_i, _i]),
    "a3vt_posenc_mask_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "a3vt_posenc_mask_bwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "a3vt_posenc_wide_supported": (_i, [_i]),
    "a3vt_posenc_wide_acts_bytes": (_sz, [_i, _i]),
    "a3vt_posenc_wide_scratch_bytes": (_sz, [_i, _i, _i]),
    "a3vt_posenc_wide_fwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "a3vt_posenc_wide_bwd": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "a3vt_image_pool_fwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "a3vt_image_pool_fwd_add": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "a3vt_image_pool_bwd": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "a3vt_bias_grad_scratch_bytes": (_sz, [ctypes.c_longlong, _i]),
    "a3vt_bias_grad_nhwc": (_i, [_vp, _i, ctypes.c_longlong, _i, _vp, _vp, _sz, _vp]),
    "a3vt_cast_weights_bf16": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_conv5_supported": (_i, [_i, _i, _i]),
    "a3vt_conv5_image_bytes": (_sz, [_i, _i, _i]),
    "a3vt_conv5_weight_image": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "a3vt_conv5_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "a3vt_conv5_input_grad_3x16s2": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "a3vt_conv5_wrw_scratch_bytes": (_sz, [_i, _i]),
    "a3vt_conv5_weight_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "a3vt_adam_chunk_elems": (_i, []),
    "a3vt_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                       ctypes.c_longlong, _vp]),
    "a3vt_bnrelu_scratch_bytes": (_sz, [_i]),
    "a3vt_bnrelu_fwd": (_i, [_vp, ctypes.c_longlong, _i, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "a3vt_bnrelu_bwd": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "a3vt_vertex_update": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "a3vt_face_cdf": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "a3vt_sample_points_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _u64, _u64,
                                    _vp, _vp, _vp, _vp, _vp]),
    "a3vt_sample_points_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_chamfer_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "a3vt_chamfer_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_chamfer_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "a3vt_chamfer_fwd_ws": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "a3vt_chamfer_fwd_shared": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    "a3vt_chamfer_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "a3vt_dbg_csr_algo": (_i, [_i]),
    "a3vt_dbg_path_counts": (_i, [_vp, _i, _i]),
    "a3vt_dbg_nn_work": (_i, [_i, _vp]),
    "a3vt_split3_bf16": (_i, [_vp, _sz, _vp, _vp, _vp, _vp]),
    "a3vt_check_finite": (_i, [_vp, _sz, _vp, _vp]),
    "a3vt_profile_enable": (_i, [_i]),
    "a3vt_profile_read": (_i, [_vp, _vp]),
    "a3vt_profile_read_classes": (_i, [_vp, _vp, _i]),
}

_LIB = None


def build(force=False, verbose=False, defines=(), out=None):
    """hipcc --offload-arch=gfx950 -shared → liba3vt.so next to this file (cross-compiles without a GPU).

    ``defines`` / ``out``: developer variants (tools/build_variants.sh) — the same per-file flags as the shipped
    library plus ``-D`` switches, written to another path (always rebuilt, objects in their own directory)."""
    if out is not None:
        return _build(True, verbose, list(defines), out, os.path.join(_HERE, "build", "variant_" + os.path.basename(out)))
    return _build(force, verbose, [], LIB_PATH, os.path.join(_HERE, "build"))


def _build(force, verbose, defines, lib_path, objdir):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("common.h", "kernels.h", "gemm_tile.h")] + \
        [os.path.join(_HERE, "..", "include", "a3vt.h")]
    force = force or os.environ.get("A3VT_FORCE_BUILD", "0") not in ("", "0")   # prove on any box that it compiles
    if not force and os.path.exists(lib_path) and all(
            os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in deps if os.path.exists(d)):
        return lib_path
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    common = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", *[f"-D{d}" for d in defines]]
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for name in SOURCES:
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        cmd = [hipcc, *common, *EXTRA_FLAGS.get(name, []), "-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd), obj))
    objs = []
    for cmd, pr, obj in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib_path]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib_path


def load():
    """Load the library (once).  Raises RuntimeError when it has not been built — no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"a3vt: native library {LIB_PATH} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        # torch first: its wheel bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  The process must
        # hold ONE HIP runtime, and it has to be the one torch initialises, or device pointers / streams handed
        # over from torch mean nothing here ("no ROCm-capable device is detected" otherwise).
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = load().a3vt_last_error()
        raise RuntimeError(f"a3vt: {what} failed (rc={rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
