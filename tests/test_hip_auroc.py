"""AUROC parity on synthetic data: 256 synthetic clips, labels Bernoulli(0.5) seed 7,
p(real) = softmax(logits)[:, 1] from the CPU oracle (fp32, = the reference's arithmetic) and from the
HIP build on identical inputs and weights; AUROC with the reference's dummy [0, 1] pair appended
(inference.py:159-160).  Bars: |dAUROC| <= 0.1 points (0.001 on the 0-1 scale) for the fp32 path
and for the bf16 path, plus rank / linear agreement of p(real), because with random weights AUROC
itself sits near 0.5."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu
from tests.cases import build_case, oracle_kwargs

pytestmark = pytest.mark.gpu


def _auroc(y, s):
    from dfd_clip_amd.harness import binary_auroc
    return binary_auroc(list(y) + [0, 1], list(s) + [0.0, 1.0])


def _spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
    return np.corrcoef(ra, rb)[0, 1]


@pytest.mark.parametrize("name,n_clips", [("small", 256), ("vitb16_cfg1", 16)])
def test_auroc_parity_on_synthetic_clips(name, n_clips):
    from dfd_clip_amd.detector import Detector
    case = build_case(name)
    T, res = case["T"], case["res"]
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.standard_normal((n_clips, T, 3, res, res), dtype=np.float32))
    m = torch.ones(n_clips, T, dtype=torch.bool)
    m[1::5, T - 1:] = False  # some padded tails
    y = np.random.default_rng(7).integers(0, 2, n_clips)
    kw = oracle_kwargs(case)
    p_ref = []
    with torch.no_grad():
        for i in range(0, n_clips, 16):
            logits, _ = ref_cpu.detector_predict(case["sd"], x[i:i + 16], m[i:i + 16], **kw)
            p_ref.append(logits[0].softmax(dim=-1)[:, 1])
    p_ref = torch.cat(p_ref).numpy()
    a_ref = _auroc(y, p_ref)
    for precision, tol_p in (("fp32", 1e-4), ("bf16", 2e-2)):
        det = Detector(case["cfg"], T, None, precision=precision)
        det.load_state_dict(case["sd"])
        det = det.to("cuda").eval()
        p = []
        with torch.no_grad():
            for i in range(0, n_clips, 32):  # ragged chunks as inference.py does
                logits, _ = det.predict(x[i:i + 32].cuda(), m[i:i + 32].cuda())
                p.append(logits[0].softmax(dim=-1)[:, 1].cpu())
        p = torch.cat(p).numpy()
        a = _auroc(y, p)
        print(f"{name}/{precision}: AUROC ref {a_ref:.4f} build {a:.4f}  max|dp| {np.abs(p - p_ref).max():.2e}  "
              f"spearman {_spearman(p, p_ref):.5f}  pearson {np.corrcoef(p, p_ref)[0, 1]:.6f}")
        assert abs(a - a_ref) <= 1e-3
        assert np.abs(p - p_ref).max() <= tol_p
        if n_clips >= 64:
            assert _spearman(p, p_ref) > 0.995
