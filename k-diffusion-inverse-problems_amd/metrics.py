"""Evaluation metrics of the caller harness -- `compute_metrics` / `calculate_average_metric`
(sample_condition_openai.py:41-68).

PSNR and SSIM are restated from their published definitions and PINNED (round 5) against scikit-image 0.18.3 itself
(`skimage.metrics.peak_signal_noise_ratio` / `structural_similarity`, the package found in this image's conda interpreter:
oracle/make_golden_thirdparty.py -> tests/golden/thirdparty_pins.npz; tests/test_thirdparty_pins.py: PSNR to 1e-6 dB, SSIM to 1e-10).  LPIPS (kdip_amd.lpips.LPIPS, the VGG forward on the
HIP conv kernels) is reported when a loaded `loss_fn_vgg` is passed, as in the reference; its pretrained
weights are not obtainable offline, so without a checkpoint the key is omitted rather than faked.
"""
import torch
import torch.nn.functional as F


def to_eval(x):
    """first sample, [-1,1] -> [0,1] (sample_condition_openai.py:42-43)."""
    return (x[0] / 2 + 0.5).clip(0, 1).detach()


def peak_signal_noise_ratio(a, b, data_range=1.0):
    mse = ((a.double() - b.double()) ** 2).mean()
    return float(10 * torch.log10(data_range ** 2 / mse))


def structural_similarity(a, b, data_range=1.0, win_size=7, K1=0.01, K2=0.03):
    """skimage.metrics.structural_similarity(im1, im2, channel_axis=0, data_range=1) defaults:
    7x7 uniform window, sample covariance (NP / (NP - 1)), mean over the (win-1)/2-cropped map,
    averaged over channels.  a, b: [C,H,W]."""
    a, b = a.double()[None], b.double()[None]
    NP = win_size * win_size
    cov_norm = NP / (NP - 1.0)
    f = lambda t: F.avg_pool2d(t, win_size, stride=1)        # valid region == the cropped uniform_filter output
    ux, uy = f(a), f(b)
    vx = cov_norm * (f(a * a) - ux * ux)
    vy = cov_norm * (f(b * b) - uy * uy)
    vxy = cov_norm * (f(a * b) - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return float(S.mean())


def compute_metrics(hat_x0, x0, loss_fn_vgg=None):
    """{'psnr', 'ssim'[, 'lpips']} of the FIRST sample against the ground truth (sample_condition_openai.py:41-49; lpips is
    evaluated on the [0,1] images exactly as the reference calls it)."""
    a, b = to_eval(x0).cpu(), to_eval(hat_x0).cpu()
    m = {"psnr": peak_signal_noise_ratio(a, b, 1.0), "ssim": structural_similarity(a, b, 1.0)}
    if loss_fn_vgg is not None:
        m["lpips"] = float(loss_fn_vgg(to_eval(x0), to_eval(hat_x0))[0, 0, 0, 0])
    return m


def calculate_average_metric(metrics_list):
    avg, cnt = {}, {}
    for m in metrics_list:
        for k, v in m.items():
            avg[k] = avg.get(k, 0.0) + v
            cnt[k] = cnt.get(k, 0) + 1
    return {k: avg[k] / cnt[k] for k in avg if cnt[k] > 0}
