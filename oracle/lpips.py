"""CPU restatement of lpips.LPIPS(net='vgg') forward (lpips v0.1: lpips/lpips.py:LPIPS.forward, normalize_tensor, NetLinLayer,
spatial_average; lpips/pretrained_networks.py:vgg16 slices) -- test infrastructure for kdip_amd.lpips.  Parity unpinned at the
package boundary: neither `lpips` nor torchvision is installed here, so the restatement follows the published source; the
reference's own call site is sample_condition_openai.py:46,161."""
import torch
import torch.nn.functional as F

SLICE_ENDS = (2, 7, 14, 21, 28)      # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 in torchvision's vgg16().features
CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)
POOL_BEFORE = (5, 10, 17, 24)


def lpips_vgg(sd, in0, in1):
    shift = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
    scale = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)
    if in0.dim() == 3:
        in0, in1 = in0[None], in1[None]

    def feats(x):
        x = (x - shift) / scale
        out = []
        for idx in CONV_IDX:
            sl = 1 + sum(idx > e for e in SLICE_ENDS)
            if idx in POOL_BEFORE:
                x = F.max_pool2d(x, 2, 2)
            x = F.relu(F.conv2d(x, sd[f"net.slice{sl}.{idx}.weight"], sd[f"net.slice{sl}.{idx}.bias"], padding=1))
            if idx in SLICE_ENDS:
                out.append(x)
        return out
    val = 0
    for k, (a, b) in enumerate(zip(feats(in0), feats(in1))):
        na = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        d = (na - nb) ** 2
        val = val + F.conv2d(d, sd[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return val
