"""CPU tests (no GPU, no compute): the C ABI's argument checks return an error code and a message instead of touching memory
— every call below passes NULL / misaligned / out-of-range arguments, fake device addresses that must never be dereferenced on
the host, or host arrays of EXACTLY the size the call may write.  ``tools/asan_host.sh`` runs this file against a build of
the library whose host side is compiled with AddressSanitizer (CPU only), so an entry point that reads or writes past one of
these arrays is caught (verdict r05 #9)."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from a3vt_amd import lib
    return lib.load()


def _err(L):
    return L.a3vt_last_error().decode()


FAKE = ctypes.c_void_p(0x7F0000001000)       # a "device pointer": 16-byte aligned, never mapped on the host
FAKE_ODD = ctypes.c_void_p(0x7F0000001008)   # 8 mod 16


def test_counters_write_exactly_n_entries(L):
    for n in (0, 1, 3, 12, 40):
        buf = (ctypes.c_longlong * max(n, 1))()
        kept = L.a3vt_dbg_path_counts(buf, n, 0)
        assert kept >= 12
    tot, cnt = (ctypes.c_double * 3)(), (ctypes.c_int * 3)()
    assert L.a3vt_profile_read_classes(tot, cnt, 3) >= 0         # (the number of classes the library keeps)
    assert L.a3vt_profile_read_classes(None, cnt, 3) < 0
    work = (ctypes.c_ulonglong * 8)()
    L.a3vt_dbg_nn_work(0, work)                                   # (reads device counters: an error code without a GPU, never a crash)


def test_csr_and_split_validators_on_host_arrays(L):
    from a3vt_amd import mesh as amesh
    v, f = amesh.icosphere(2)
    csr = amesh.CSRAdjacency.from_pairs(*amesh.vision_pairs(f, v.shape[0]), v.shape[0])
    rp = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
    col = np.ascontiguousarray(csr.col, dtype=np.int32)
    assert L.a3vt_csr_validate(rp.ctypes.data, col.ctypes.data, v.shape[0], col.size) == 0
    bad = col.copy()
    bad[-1] = v.shape[0]                                   # a column index one past the end
    assert L.a3vt_csr_validate(rp.ctypes.data, bad.ctypes.data, v.shape[0], col.size) != 0
    assert L.a3vt_csr_validate(rp.ctypes.data, col.ctypes.data, v.shape[0], col.size - 1) != 0    # nnz disagrees with rowptr
    assert L.a3vt_csr_validate(None, col.ctypes.data, v.shape[0], col.size) != 0
    rp2 = rp.copy()
    rp2[3] = rp2[2] - 1                                    # not monotone
    assert L.a3vt_csr_validate(rp2.ctypes.data, col.ctypes.data, v.shape[0], col.size) != 0


def test_size_queries_never_fail_on_odd_sizes(L):
    for b, n, i, h, nl, c in ((1, 1, 1, 4, 1, 0), (64, 2562, 50, 300, 20, 99), (8, 10242, 448, 300, 20, 99), (3, 17, 600, 304, 2, 304),
                              (0, 0, 0, 0, 0, 0), (-1, 5, 5, 5, 5, 5)):
        for mode in (0, 1, 2, 3):
            L.a3vt_gcn_stack_scratch_bytes_mode(b, n, i, h, nl, c, 1, mode)
        L.a3vt_gcn_stack_scratch_bytes(b, n, i, h, nl, c, 1)
        L.a3vt_gcn_stack_mask_bytes(b, n, h, nl, c)
        ab, mb = ctypes.c_size_t(), ctypes.c_size_t()
        L.a3vt_gcn_stack_stash_bytes(b, n, h, nl, c, 2, ctypes.byref(ab), ctypes.byref(mb))
    assert L.a3vt_gcn_stack_stash_bytes(2, 10, 300, 3, 99, 0, None, None) != 0
    assert L.a3vt_bnrelu_scratch_bytes(0) == 0 and L.a3vt_bnrelu_scratch_bytes(16) > 0
    assert L.a3vt_bias_grad_scratch_bytes(0, 16) == 0
    L.a3vt_chamfer_workspace_bytes(3, 64, 10000, 10000)
    L.a3vt_chamfer_workspace_bytes(0, 0, 0, 0)


def test_entry_points_refuse_bad_arguments_before_touching_them(L):
    f32 = ctypes.c_float
    # BatchNorm + ReLU: fewer than two rows, misaligned maps, one running statistic without the other, short scratch
    need = L.a3vt_bnrelu_scratch_bytes(16)
    args = lambda **kw: [kw.get("x", FAKE), kw.get("rows", 100), kw.get("c", 16), FAKE, FAKE, None, f32(1e-5), f32(0.1),   # noqa: E731
                         kw.get("rm", FAKE), kw.get("rv", FAKE), None, kw.get("y", FAKE), FAKE, FAKE, kw.get("sb", need), None]
    assert L.a3vt_bnrelu_fwd(*args(rows=1)) != 0
    assert L.a3vt_bnrelu_fwd(*args(x=FAKE_ODD)) != 0
    assert L.a3vt_bnrelu_fwd(*args(rv=None)) != 0
    assert L.a3vt_bnrelu_fwd(*args(sb=need - 1)) != 0
    assert L.a3vt_bnrelu_fwd(*args(c=0)) != 0
    assert L.a3vt_bnrelu_bwd(FAKE, None, 100, 16, FAKE, FAKE, FAKE, FAKE, None, FAKE, need, None) != 0
    # batched weight cast: too many tensors; a NULL entry among host arrays of exactly n entries
    n = 3
    src = (ctypes.c_void_p * n)(FAKE.value, None, FAKE.value)
    dst = (ctypes.c_void_p * n)(FAKE.value, FAKE.value, FAKE.value)
    outer = (ctypes.c_longlong * n)(4, 4, 4)
    inner = (ctypes.c_int * n)(3, 3, 3)
    hw = (ctypes.c_int * n)(25, 25, 25)
    assert L.a3vt_cast_weights_bf16(n, src, dst, outer, inner, hw, None) != 0
    assert L.a3vt_cast_weights_bf16(97, src, dst, outer, inner, hw, None) != 0
    assert L.a3vt_cast_weights_bf16(0, None, None, None, None, None, None) == 0
    # bias gradient, stack calls: NULL operands
    assert L.a3vt_bias_grad_nhwc(None, 1, 100, 16, FAKE, FAKE, 1 << 20, None) != 0
    assert L.a3vt_bias_grad_nhwc(FAKE_ODD, 1, 100, 16, FAKE, FAKE, 1 << 20, None) != 0
    assert L.a3vt_gcn_stack_fwd(None, 52, 50, None, None, 20, 300, 99, None, None, None, 7, 2562, 64, 0, None, None, None, None, None) != 0
    assert _err(L) != ""
    assert L.a3vt_gcn_stack_fwd(FAKE, 51, 50, FAKE, FAKE, 20, 300, 99, FAKE, FAKE, FAKE, 7, 2562, 64, 7, None, None, FAKE, FAKE, None) != 0   # mode 7
    assert L.a3vt_dbg_csr_algo(9) != 0


def test_conv5_and_adam_entry_points_refuse_bad_arguments(L):
    """Round 6 entry points: the direct convolution's weight gradient and the one-launch optimizer step."""
    # shapes: only the five layer shapes of the pyramid; scratch sizes are 0 for everything else
    assert L.a3vt_conv5_supported(16, 16, 1) == 1 and L.a3vt_conv5_supported(3, 16, 2) == 1 and L.a3vt_conv5_supported(64, 64, 1) == 0
    for cin, cout in ((3, 3), (3, 16), (16, 16), (16, 32), (32, 32)):
        assert L.a3vt_conv5_wrw_scratch_bytes(cin, cout) > 0
    assert L.a3vt_conv5_wrw_scratch_bytes(64, 64) == 0 and L.a3vt_conv5_wrw_scratch_bytes(0, 0) == 0
    need = L.a3vt_conv5_wrw_scratch_bytes(16, 16)
    wg = lambda **kw: [kw.get("x", FAKE), kw.get("gy", FAKE), kw.get("b", 2), kw.get("h", 20), kw.get("w", 20), kw.get("cin", 16),   # noqa: E731
                       kw.get("cout", 16), kw.get("s", 1), kw.get("gw", FAKE), kw.get("scr", FAKE), kw.get("sb", need), None]
    assert L.a3vt_conv5_weight_grad(*wg(x=None)) != 0
    assert L.a3vt_conv5_weight_grad(*wg(gw=None)) != 0
    assert L.a3vt_conv5_weight_grad(*wg(x=FAKE_ODD)) != 0            # 16-channel maps: 16-byte aligned
    assert L.a3vt_conv5_weight_grad(*wg(scr=FAKE_ODD)) != 0
    assert L.a3vt_conv5_weight_grad(*wg(sb=need - 1)) != 0
    assert L.a3vt_conv5_weight_grad(*wg(cin=64, cout=64)) != 0
    assert L.a3vt_conv5_weight_grad(*wg(s=2)) != 0                   # 16 -> 16 exists at stride 1 only
    assert L.a3vt_conv5_weight_grad(*wg(h=2)) != 0 and L.a3vt_conv5_weight_grad(*wg(b=0)) != 0
    assert "argument check failed" in _err(L)
    # Adam: an empty table is a no-op; NULL tables, step 0 and hyper-parameters outside their ranges are refused
    d = ctypes.c_double
    ad = lambda **kw: [kw.get("p", FAKE), FAKE, FAKE, FAKE, FAKE, FAKE, FAKE, kw.get("n", 4), d(kw.get("lr", 3e-4)), d(kw.get("b1", 0.9)),   # noqa: E731
                       d(kw.get("b2", 0.999)), d(1e-8), d(0.0), kw.get("step", 1), None]
    assert L.a3vt_adam_step(None, None, None, None, None, None, None, 0, d(3e-4), d(0.9), d(0.999), d(1e-8), d(0.0), 1, None) == 0
    assert L.a3vt_adam_step(*ad(p=None)) != 0
    assert L.a3vt_adam_step(*ad(step=0)) != 0
    assert L.a3vt_adam_step(*ad(b1=1.0)) != 0 and L.a3vt_adam_step(*ad(b2=-0.1)) != 0 and L.a3vt_adam_step(*ad(lr=-1.0)) != 0
    assert L.a3vt_adam_step(*ad(n=-1)) != 0
    assert L.a3vt_adam_chunk_elems() == 4096
