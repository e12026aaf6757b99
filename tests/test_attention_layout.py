"""The data flow of attn_fwd_kernel (comorag_amd/csrc/encoder_kernels.hip) replayed in numpy, lane by lane: MFMA operand / accumulator
layouts (cmr_device.h: lane l of an A or B operand holds row / column l & 31 and k = 8*(l >> 5) + [0,8); accumulator register r of
lane l is row (r & 3) + 8*(r >> 2) + 4*(l >> 5), column l & 31), the transposed scores, the k-slot permutation that lets the
probabilities feed the second MFMA from the registers they were computed in, the V^T image in LDS, masking and the online softmax.
No GPU: this pins the index arithmetic DESIGN.md §4.6 describes; the kernel itself is tested in tests/test_encoder_fused_gpu.py."""
import numpy as np


def acc_row(r, lane): return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
def mma(a, b, c):
    # a,b: [64 lanes][8], c: [64][16]
    A = np.zeros((32, 16)); B = np.zeros((32, 16))
    for l in range(64):
        for e in range(8):
            A[l & 31, 8 * (l >> 5) + e] = a[l][e]
            B[l & 31, 8 * (l >> 5) + e] = b[l][e]
    D = A @ B.T
    out = c.copy()
    for l in range(64):
        for r in range(16):
            out[l][r] += D[acc_row(r, l), l & 31]
    return out
L = 100; length = 77; hidden = 128; nh = 2; rs = 3 * hidden      # one 100-token row of a mini-batch, 77 real tokens, 2 heads
qkv = np.random.default_rng(0).standard_normal((L, rs)).astype(np.float32)
head = 1
KSTR, VSTR = 72, 68
def run_block(q0, wave):
    qbase = head * 64; kbase = hidden + head * 64; vbase = 2 * hidden + head * 64
    # Q frags
    qf = np.zeros((4, 64, 8))
    for lane in range(64):
        g, c = lane >> 5, lane & 31
        qrow = q0 + wave * 32 + c
        for ks in range(4):
            if qrow < L: qf[ks, lane] = qkv[qrow, qbase + ks * 16 + g * 8: qbase + ks * 16 + g * 8 + 8]
    o = np.zeros((2, 64, 16)); m = np.full(64, -1e30); lsum = np.zeros(64)
    nch = (length + 63) // 64
    sc = 1.4426950408889634 * 0.125
    for ch in range(nch):
        k_lds = np.zeros(64 * KSTR); v_lds = np.zeros(64 * VSTR)
        for tid in range(256):
            lane, w = tid & 63, tid >> 6
            kr, kseg = tid >> 3, tid & 7
            pi, dseg = w * 8 + (lane & 7), lane >> 3
            kreg = np.zeros((2, 8)); vreg = np.zeros((2, 8))
            for h in range(2):
                key = ch * 64 + kr + 32 * h
                if key < length: kreg[h] = qkv[key, kbase + kseg * 8: kbase + kseg * 8 + 8]
                vkey = ch * 64 + 2 * pi + h
                if vkey < length: vreg[h] = qkv[vkey, vbase + dseg * 8: vbase + dseg * 8 + 8]
            for h in range(2):
                k_lds[(kr + 32 * h) * KSTR + kseg * 8:(kr + 32 * h) * KSTR + kseg * 8 + 8] = kreg[h]
            for wd in range(4):
                # dword = (a & 0xffff) | (b << 16): low = a elem 2wd, high = b elem 2wd
                base = (dseg * 8 + 2 * wd) * VSTR + 2 * pi
                v_lds[base] = vreg[0][2 * wd]; v_lds[base + 1] = vreg[1][2 * wd]
                base = (dseg * 8 + 2 * wd + 1) * VSTR + 2 * pi
                v_lds[base] = vreg[0][2 * wd + 1]; v_lds[base + 1] = vreg[1][2 * wd + 1]
        s = np.zeros((2, 64, 16))
        for t in range(2):
            for ks in range(4):
                kf = np.zeros((64, 8))
                for lane in range(64):
                    g, c = lane >> 5, lane & 31
                    off = (t * 32 + c) * KSTR + ks * 16 + g * 8
                    kf[lane] = k_lds[off:off + 8]
                s[t] = mma(kf, qf[ks], s[t])
        k0 = ch * 64
        for t in range(2):
            for lane in range(64):
                for r in range(16):
                    if k0 + t * 32 + acc_row(r, lane) >= length: s[t, lane, r] = -1e30
        cm = s.max(axis=(0, 2))
        cm = np.maximum(cm, cm[np.arange(64) ^ 32])
        mn = np.maximum(m, cm)
        alpha = np.exp2((m - mn) * sc)
        p = np.exp2(s * sc - (mn * sc)[None, :, None])
        ps = p.sum(axis=(0, 2))
        lsum = lsum * alpha + ps; m = mn
        o *= alpha[None, :, None]
        for t in range(2):
            for sp in range(2):
                pb = p[t][:, 8 * sp: 8 * sp + 8]
                for dt in range(2):
                    va = np.zeros((64, 8))
                    for lane in range(64):
                        g, c = lane >> 5, lane & 31
                        kb = t * 32 + sp * 16 + 4 * g
                        off = (dt * 32 + c) * VSTR + kb
                        va[lane, :4] = v_lds[off:off + 4]; va[lane, 4:] = v_lds[off + 8: off + 12]
                    o[dt] = mma(va, pb, o[dt])
    inv = 1.0 / (lsum + lsum[np.arange(64) ^ 32])
    out = np.full((32, 64), np.nan)
    for lane in range(64):
        g, c = lane >> 5, lane & 31
        for dt in range(2):
            for rr in range(4):
                for j in range(4):
                    out[c, dt * 32 + 8 * rr + 4 * g + j] = o[dt, lane, 4 * rr + j] * inv[lane]
    return out


def test_attention_kernel_data_flow_equals_softmax_attention():
    Q = qkv[:, head * 64: head * 64 + 64]
    K = qkv[:length, hidden + head * 64: hidden + head * 64 + 64]
    V = qkv[:length, 2 * hidden + head * 64: 2 * hidden + head * 64 + 64]
    S = Q @ K.T / 8
    P = np.exp(S - S.max(1, keepdims=True))
    P /= P.sum(1, keepdims=True)
    ref = P @ V
    for q0, wave in ((0, 0), (0, 2), (0, 3)):          # rows 0-31, 64-95, 96-99 (+ 28 rows past the end of the mini-batch)
        got = run_block(q0, wave)
        rows = np.arange(q0 + wave * 32, min(q0 + wave * 32 + 32, L))
        assert np.abs(got[:len(rows)] - ref[rows]).max() < 1e-5
