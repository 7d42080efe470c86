"""Independent pins for the oracle rows the reference itself has no vectors for: PyTorch (CPU)
is the framework the shipped models were trained and exported from (misc/taiyaki_flipflop5_guppy.py:77-99 writes
torch.nn.LSTM's weight_ih_l0 / weight_hh_l0 / bias_ih_l0 and Conv1d weights into the .mdl layout), so
  * the oracle's lstm_forward/backward must equal torch.nn.LSTM run on those tensors (gate order i,f,g,o),
  * its convolution must equal torch conv1d wherever the reference's window is complete (stride 1: everywhere),
  * its transition posterior must equal the gradient of the log partition function with respect to the transition
    scores (an identity of the CRF, evaluated here by autograd in fp64), and logZ itself a torch.logsumexp recursion.
CPU only; torch is test infrastructure here, never part of the product."""
import numpy as np
import pytest
import torch

from oracle import ffo

torch.set_num_threads(1)


def omat(a):
    return ffo.HostMat.from_dense(np.ascontiguousarray(a, dtype=np.float32))


@pytest.mark.parametrize("H,T", [(8, 11), (32, 40), (96, 25)])
def test_lstm_equals_torch_lstm(H, T):
    torch.manual_seed(H + T)
    lstm = torch.nn.LSTM(H, H, bias=True)
    with torch.no_grad():
        lstm.bias_hh_l0.zero_()                                     # the export keeps bias_ih only (taiyaki_flipflop5_guppy.py:80)
    x = torch.randn(T, 1, H)
    lib = ffo.lib()
    # flappie layout: iW [H x 4H] stored one column per gate row = weight_ih_l0 [4H, H] row by row (:77-79)
    iW, sW = lstm.weight_ih_l0.detach().numpy(), lstm.weight_hh_l0.detach().numpy()
    b = lstm.bias_ih_l0.detach().numpy()[None, :]
    xin = x[:, 0, :].numpy()
    for backward in (0, 1):
        xa = ffo.take(lib.fo_affine_map(omat(xin).ptr, omat(iW).ptr, omat(b).ptr))                 # [T, 4H]
        got = ffo.take(lib.fo_lstm(omat(xa).ptr, omat(sW).ptr, backward))                             # [T, H]
        with torch.no_grad():
            xt = torch.flip(x, [0]) if backward else x
            want, _ = lstm(xt)
            want = (torch.flip(want, [0]) if backward else want)[:, 0, :].numpy()
        np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)


@pytest.mark.parametrize("nf,nfilter,winlen,stride,T", [(1, 4, 5, 1, 50), (4, 16, 5, 1, 33), (16, 24, 19, 5, 203), (16, 24, 19, 5, 200),
                                                          (1, 8, 19, 2, 77)])
def test_convolution_equals_torch_conv1d_where_windows_are_complete(nf, nfilter, winlen, stride, T):
    rng = np.random.default_rng(winlen + T)
    x = rng.standard_normal((T, nf)).astype(np.float32)
    wt = (rng.standard_normal((nfilter, nf, winlen)) / np.sqrt(nf * winlen)).astype(np.float32)     # torch Conv1d weight [out, in, k]
    b = rng.standard_normal(nfilter).astype(np.float32)
    # .mdl layout (taiyaki_flipflop5_guppy.py:85-94): column per filter, rows (tap, feature padded to 4)
    nfp = 4 * ((nf + 3) // 4)
    w = np.zeros((nfilter, winlen * nfp), dtype=np.float32)
    for t in range(winlen):
        w[:, t * nfp:t * nfp + nf] = wt[:, :, t]
    w = w[:, :nfp * winlen - nfp + nf]
    got = ffo.take(ffo.lib().fo_convolution(omat(x).ptr, omat(w).ptr, omat(b[None, :]).ptr, stride))    # [Tout, nfilter]
    padL, padR = (winlen - 1) // 2, winlen // 2
    xt = torch.nn.functional.pad(torch.from_numpy(x.T[None]), (padL, padR))
    want = torch.nn.functional.conv1d(xt, torch.from_numpy(wt), torch.from_numpy(b), stride=stride)[0].numpy().T
    assert got.shape[0] == (T + stride - 1) // stride
    n = min(got.shape[0], want.shape[0])
    if stride == 1:
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)                                     # zero-padded "same" convolution
    else:
        # the reference's strided right edge is its own (partial windows there are even
        # accumulated into columns that already hold a complete one); away from it, columns whose window lies inside
        # [0, T) must agree with the textbook strided convolution
        tail = (winlen + stride - 1) // stride + 1
        ok = [c for c in range(n - tail) if c * stride - padL >= 0 and c * stride - padL + winlen <= T]
        assert len(ok) >= n // 2
        np.testing.assert_allclose(got[ok], want[ok], rtol=0, atol=1e-5)              # 304-term fp32 sums, different order


def torch_logz(S, nbase):
    """log partition function of a flip-flop CRF, scores S [T, P] (fp64), initial vector zero (layers.c:1035-1079)"""
    ns = 2 * nbase
    alpha = torch.zeros(ns, dtype=S.dtype)
    for t in range(S.shape[0]):
        flip = S[t, :nbase * ns].reshape(nbase, ns)                          # [to, from]
        new_flip = torch.logsumexp(flip + alpha[None, :], dim=1)
        stay, move = S[t, nbase * ns + nbase:nbase * ns + ns], S[t, nbase * ns:nbase * ns + nbase]
        new_flop = torch.logsumexp(torch.stack([alpha[nbase:] + stay, alpha[:nbase] + move]), dim=0)
        alpha = torch.cat([new_flip, new_flop])
    return torch.logsumexp(alpha, dim=0)


@pytest.mark.parametrize("nbase,T", [(4, 60), (5, 33), (2, 7)])
def test_partition_function_and_posterior_against_autograd(nbase, T):
    P = 2 * nbase * (nbase + 1)
    rng = np.random.default_rng(nbase * T)
    s = (rng.standard_normal((T, P)) * 1.5).astype(np.float32)
    S = torch.tensor(s, dtype=torch.float64, requires_grad=True)
    logz = torch_logz(S, nbase)
    lib = ffo.lib()
    assert abs(lib.fo_partition_function(omat(s).ptr) - float(logz)) <= 1e-9 * max(1.0, abs(float(logz)))
    logz.backward()
    want = S.grad.numpy()                                                     # posterior probability of each transition
    got = ffo.take(lib.fo_transpost(omat(s).ptr, 0))                          # transpost_crf_flipflop(trans, return_log = false)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)              # fp32 log-space chains against exact fp64
    np.testing.assert_allclose(got.sum(axis=1), 1.0, rtol=0, atol=1e-4)
