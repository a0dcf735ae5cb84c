"""TEST INFRASTRUCTURE (oracle): the eval-image metrics of FruitModel.get_image_metrics_and_images
(/root/reference/fruit_nerf/fruit_nerf.py:403-458) restated in float64 NumPy.

torchmetrics is a third-party dependency that is absent here (not vendored in /root/reference, not installable: parity
"recalled" for its constants, like nerfstudio's); its published algorithm for the defaults the reference uses
(`structural_similarity_index_measure(preds, target, gaussian_kernel=True, sigma=1.5, kernel_size=11, data_range=None,
k1=0.01, k2=0.03)`, torchmetrics/functional/image/ssim.py; the reference passes no data_range — fruit_nerf.py:176,424 —
so torchmetrics' `_ssim_update` takes `max(preds.max() - preds.min(), target.max() - target.min())`):
    window w = outer(g, g), g_i = exp(-(d_i / sigma)^2 / 2) / sum, d = arange((1 - 11) / 2, (1 + 11) / 2)  (float32 in
    torchmetrics: the window is rounded to float32 here too, everything after that is float64);
    reflect-pad preds and target by 5; mu_p, mu_t, E[pp], E[tt], E[pt] = valid 11 x 11 filtering of the padded images;
    sigma_p^2 = E[pp] - mu_p^2, sigma_t^2 = E[tt] - mu_t^2, sigma_pt = E[pt] - mu_p mu_t;
    ssim = ((2 mu_p mu_t + c1)(2 sigma_pt + c2)) / ((mu_p^2 + mu_t^2 + c1)(sigma_p^2 + sigma_t^2 + c2)), c1 = (0.01 R)^2, c2 = (0.03 R)^2 with R the data range;
    the map is cropped by the pad on every side again and averaged — so only windows that never touch the padding count.
PSNR(data_range = 1) = 10 log10(1 / mse); BinaryJaccardIndex(threshold = 0.5) = |pred & target| / |pred | target|
(0 for an empty union).  The reference's `F.softmax(outputs["semantics"])` has no dim: torch's legacy implicit dim for a
3-D tensor is 0 (torch.nn.functional._get_softmax_dim), a softmax over image rows.
Only tests/, bench.py's checker legs and __graft_entry__.smoke() may import this module."""
import numpy as np


def gaussian_window(kernel_size: int = 11, sigma: float = 1.5) -> np.ndarray:
    dist = np.arange((1 - kernel_size) / 2, (1 + kernel_size) / 2, 1, dtype=np.float32)
    g = np.exp(-((dist / np.float32(sigma)) ** 2) / 2).astype(np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def ssim(pred: np.ndarray, target: np.ndarray, kernel_size: int = 11, sigma: float = 1.5, k1: float = 0.01,
         k2: float = 0.03, data_range=None) -> float:
    """pred, target: [H, W, C] -> mean SSIM (float64).  data_range None (torchmetrics' default, what the reference uses):
    the larger of the two images' value ranges."""
    p = np.moveaxis(np.asarray(pred, np.float64), -1, 0)
    t = np.moveaxis(np.asarray(target, np.float64), -1, 0)
    if data_range is None:
        data_range = max(float(p.max() - p.min()), float(t.max() - t.min()))
    g = gaussian_window(kernel_size, sigma).astype(np.float64)
    pad = (kernel_size - 1) // 2
    p = np.pad(p, ((0, 0), (pad, pad), (pad, pad)), mode="reflect")
    t = np.pad(t, ((0, 0), (pad, pad), (pad, pad)), mode="reflect")

    def filt(x):   # valid 2-D filtering with outer(g, g), rows then columns (exact in any order up to float64 rounding)
        H, W = x.shape[1] - 2 * pad, x.shape[2] - 2 * pad
        rows = sum(g[k] * x[:, :, k:k + W] for k in range(kernel_size))
        return sum(g[k] * rows[:, k:k + H, :] for k in range(kernel_size))

    mu_p, mu_t = filt(p), filt(t)
    s_pp, s_tt, s_pt = filt(p * p) - mu_p * mu_p, filt(t * t) - mu_t * mu_t, filt(p * t) - mu_p * mu_t
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    m = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    return float(m[:, pad:-pad, pad:-pad].mean())


def psnr(pred: np.ndarray, target: np.ndarray) -> float:
    mse = np.mean((np.asarray(pred, np.float64) - np.asarray(target, np.float64)) ** 2)
    return float(10.0 * np.log10(1.0 / mse))


def jaccard(pred: np.ndarray, target: np.ndarray) -> float:
    inter, union = np.logical_and(pred, target).sum(), np.logical_or(pred, target).sum()
    return float(inter / union) if union > 0 else 0.0


def image_metrics(rgb: np.ndarray, image: np.ndarray, semantics: np.ndarray, mask: np.ndarray) -> dict:
    """rgb (unclamped) / image [H,W,3], semantics logits / mask [H,W,1] -> the reference's metrics_dict entries (+ the
    meaningful sigmoid IoU)."""
    rgb = np.clip(np.asarray(rgb, np.float64), 0.0, 1.0)
    sem = np.asarray(semantics, np.float64)
    tgt = np.asarray(mask)[..., 0] > 0.5
    e = np.exp(sem - sem.max(axis=0, keepdims=True))
    row_softmax = e / e.sum(axis=0, keepdims=True)           # implicit dim 0 of a 3-D tensor
    return {"psnr": psnr(rgb, image), "ssim": ssim(image, rgb),
            "iou": jaccard(row_softmax[..., 0] > 0.5, tgt),
            "iou_sigmoid": jaccard(1.0 / (1.0 + np.exp(-sem[..., 0])) > 0.5, tgt)}
