"""The numerics claim behind the split-bf16 recurrent layer kernel (flappie_amd/csrc/ffhip_rnn_split.hip),
checked in numpy on the CPU: a three-way bf16 split is exact for every fp32 value, and the six products the kernel keeps
reproduce an fp32 dot product at least as well as fp32 arithmetic does; fewer terms do not."""
import numpy as np


def bf16_round(x):
    """round to nearest even to 8 mantissa bits, returned as float32 (what v_cvt_pk_bf16_f32 and the host packer do)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    a = bf16_round(x)
    r1 = x - a
    b = bf16_round(r1)
    c = bf16_round(r1 - b)
    return a, b, c


def test_three_bf16_slices_hold_any_fp32_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000), rng.uniform(-1, 1, 200000) * 1e-6, rng.uniform(-1, 1, 1000) * 1e30,
                        [0.0, 1.0, -1.0, 0.99999994, 1.17549435e-38, 3.4028235e38 / 4]]).astype(np.float32)
    a, b, c = split3(x)
    assert np.array_equal((a.astype(np.float64) + b.astype(np.float64) + c.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(a + (b + c), x)                      # also when summed in fp32, smallest first
    for s in (a, b, c):                                        # every slice really is a bf16
        assert np.all((s.view(np.uint32) & np.uint32(0xFFFF)) == 0)


def test_six_terms_match_fp32_five_do_not():
    rng = np.random.default_rng(1)
    K, M, N = 768, 512, 16                                     # one gate row block of the H = 384 layer: [Wi | sW] . [x ; h]
    W = (rng.uniform(-1, 1, (M, K)) * 3 / np.sqrt(K)).astype(np.float32)
    X = np.tanh(rng.standard_normal((K, N))).astype(np.float32)
    exact = W.astype(np.float64) @ X.astype(np.float64)
    w, x = split3(W), split3(X)

    def err(pairs):
        acc = np.zeros((M, N), dtype=np.float32)
        for i, j in pairs:                                     # each product exact in fp32, fp32 accumulation: what the MFMA does
            acc = acc + (w[i] @ x[j])
        return float(np.abs(acc - exact).max())
    six = err([(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])
    five = err([(0, 2), (1, 1), (1, 0), (0, 1), (0, 0)])
    three = err([(1, 0), (0, 1), (0, 0)])
    one = err([(0, 0)])
    fp32 = float(np.abs(W @ X - exact).max())
    assert six <= fp32                                         # products do not round; the accumulation is fp32 either way
    assert six < 3e-6 and five > 3 * six and three > 10 * six and one > 1e-3


# ---- the default operand format: two fp16 slices of the operand scaled into fp16's range (ffhip_split.hpp) -------------------
def split_f16(x, exp2):
    v = np.asarray(x, dtype=np.float32) * np.float32(2.0 ** exp2)
    a = v.astype(np.float16).astype(np.float32)
    b = (v - a).astype(np.float16).astype(np.float32)
    return a, b


def test_two_fp16_slices_hold_22_bits():
    rng = np.random.default_rng(2)
    x = np.concatenate([np.tanh(rng.standard_normal(200000)), rng.uniform(-1, 1, 100000) * 1e-3, rng.uniform(-1, 1, 100000) * 1e-6]).astype(np.float32)
    a, b = split_f16(x, 12)
    back = (a.astype(np.float64) + b.astype(np.float64)) / 4096.0
    err = np.abs(back - x.astype(np.float64))
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -37))
    assert np.all(np.abs(a) <= 4096.0)                          # in fp16's range with room to spare


def test_three_fp16_products_are_as_good_as_an_fp32_gemm():
    """the three products the default kernels keep (w0 x0, w0 x1, w1 x0): error against float64 at the level of a plain fp32 GEMM
    and of the reference's sequential fp32 dot product; the fourth product (w1 x1) buys nothing; one product alone is useless"""
    rng = np.random.default_rng(1)
    K, M, N = 768, 512, 16
    worst = 0.0
    for trial in range(3):
        W = (rng.uniform(-1, 1, (M, K)) * 3 / np.sqrt(K)).astype(np.float32)
        X = np.tanh(rng.standard_normal((K, N))).astype(np.float32)
        if trial == 1:
            X = (X * rng.choice([1, 1e-2, 1e-4], (K, N))).astype(np.float32)      # small activations: second slices in fp16's subnormal range
        exact = W.astype(np.float64) @ X.astype(np.float64)
        sw = int(np.floor(np.log2(32768 / np.abs(W).max())))
        w, x = split_f16(W, sw), split_f16(X, 12)

        def err(pairs):
            acc = np.zeros((M, N), dtype=np.float32)
            for i, j in pairs:
                acc = acc + (w[i] @ x[j])
            return float(np.abs(acc.astype(np.float64) / 2.0 ** (sw + 12) - exact).max())
        three = err([(1, 0), (0, 1), (0, 0)])
        four = err([(1, 1), (1, 0), (0, 1), (0, 0)])
        one = err([(0, 0)])
        fp32 = float(np.abs(W @ X - exact).max())
        seq = np.zeros((64, N), dtype=np.float32)                # the reference's order: term by term
        for k in range(K):
            seq = seq + W[:64, k:k + 1] * X[k:k + 1, :]
        seq_err = float(np.abs(seq - exact[:64]).max())
        assert three <= 1.25 * max(fp32, seq_err), (trial, three, fp32, seq_err)
        assert four >= 0.8 * three and one > 100 * three
        worst = max(worst, three)
    assert worst < 3e-6
