#!/usr/bin/env python3
"""Caller harness of the hot path: posterior sampling for a folder of images.

Mirrors the reference's `sample_condition_openai.py` (flag names :74-100, model / operator
construction from the same JSON / YAML keys :112-151 -- the reference's own per-model / per-task config files can be passed as
they are, `configs/models.json#<entry>` / `configs/tasks.yaml#<entry>` hold the BASELINE configurations in one file each, x_T = randn * sigma_max :187, churn
constants :191, per-image metric dict + avg_metrics.yaml :196-213) on top of the MI355X path
(`kdip_amd`).  Differences, all explicit:

  * `--v2` (or a config with `ortho_tf_type`) is the reference's second script, `sample_condition_openai_v2.py`:
    `OpenAIDenoiserV2` with the `out_cov` log-variance head, `ConditionOpenAIDenoiserV2`, `--spatial-var`,
    `--mle-sigma-thres` default 1 instead of 0.2 (:70-91,150-160).  Its Lightning checkpoint is read through
    `ku.normalize_state_dict` (`model_ema.inner_model.*` / `model_ema.out_cov.*` keys); `--synthetic-weights` otherwise.
  * `--batch-size` > 1 is allowed: B independent batch-1 problems per call (the reference asserts 1).
  * no checkpoint / dataset is obtainable offline, so `--synthetic-weights` (seeded random-init
    weights of the configured architecture) and `--synthetic-data N` (seeded smooth images) stand in
    for `--checkpoint` and the image folder; real ones are used when the paths exist.
  * metrics: psnr + ssim (kdip_amd.metrics) + lpips when `--lpips-checkpoint` names a VGG / lin state_dict (kdip_amd.lpips;
    the pretrained weights are not obtainable offline).
  * one process per GPU: launch under `python -m torch.distributed.run` for multi-GPU; the samples of
    an image are split over the ranks and all-gathered once (kdip_amd.evaluation.compute_features).
"""
import argparse
import json
import os
from functools import partial

import torch
import yaml

import kdip_amd  # noqa: F401  (alias loader for the package directory)
import kdip_amd.condition as kc
import kdip_amd.evaluation as ke
import kdip_amd.measurements as km
import kdip_amd.metrics as kmet
import kdip_amd.sampling as ks
import kdip_amd.unet as ku


def load_yaml(path):
    """`file.yaml` (one operator per file, the reference's layout) or `file.yaml#entry` (configs/tasks.yaml)."""
    path, _, entry = path.partition("#")
    with open(path) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    return cfg[entry] if entry else cfg


def load_json(path):
    """`file.json` (one model per file, the reference's layout) or `file.json#entry` (configs/models.json)."""
    path, _, entry = path.partition("#")
    cfg = json.load(open(path))
    return cfg[entry] if entry else cfg


def save_yaml(data, path):
    with open(path, "w") as f:
        yaml.dump(data, f)


def folder_of_images(root):
    """K.utils.FolderOfImages + ToTensor + (x*2-1) (sample_condition_openai.py:138-143): sorted image files -> [3,H,W] in [-1,1]."""
    import numpy as np
    from PIL import Image
    exts = {".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp"}
    paths = sorted(os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if os.path.splitext(f)[1].lower() in exts)
    for p in paths:
        img = np.asarray(Image.open(p).convert("RGB"), dtype=np.float32) / 255.0
        yield torch.from_numpy(img).permute(2, 0, 1).contiguous() * 2 - 1


def synthetic_images(n, size):
    """seeded smooth fields in [-1,1]: clamp(3 * avgpool9x9(rand*2-1, circular), -1, 1)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(1)
    for _ in range(n):
        x = torch.rand(1, 3, size, size, generator=g) * 2 - 1
        x = F.avg_pool2d(F.pad(x, (4, 4, 4, 4), mode="circular"), 9, stride=1)
        yield (3 * x).clamp(-1, 1)[0]


def to_pil_image(x):
    """K.utils.to_pil_image: [-1,1] CHW -> 8-bit RGB."""
    from PIL import Image
    a = ((x.detach().float().cpu().clamp(-1, 1) + 1) / 2 * 255).round().to(torch.uint8).permute(1, 2, 0).numpy()
    return Image.fromarray(a)


def main():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--batch-size", type=int, default=1, help="the batch size (samples per sampler call)")
    p.add_argument("--checkpoint", type=str, default="../model_zoo/diffusion_ffhq_10m.pt", help="the checkpoint to use")
    p.add_argument("--config", type=str, default="configs/models.json#ffhq", help="the model config (file.json or file.json#entry)")
    p.add_argument("--operator-config", type=str, default="configs/tasks.yaml#inpainting", help="file.yaml or file.yaml#entry")
    p.add_argument("-n", type=int, default=1, help="the number of images to sample per measurement")
    p.add_argument("--prefix", type=str, default="out", help="the output prefix")
    p.add_argument("--logdir", type=str, default=os.path.join("runs", "sample_condition", "temp"))
    p.add_argument("--save-img", dest="save_img", action="store_true")
    # sampler
    p.add_argument("--steps", type=int, default=50, help="the number of denoising steps")
    p.add_argument("--ode", dest="ode", action="store_true")
    p.add_argument("--euler", dest="euler", action="store_true")
    # guidance
    p.add_argument("--guidance", type=str, default="I")
    p.add_argument("--xstart-cov-type", type=str, choices=["analytic", "convert", "pgdm", "dps", "diffpir", "tmpd"], default="convert")
    p.add_argument("--mle-sigma-thres", type=float, default=None, help="default 0.2 (1 with --v2)")
    p.add_argument("--lam", type=float, default=None)
    p.add_argument("--zeta", type=float, default=None)
    p.add_argument("--num-hutchinson-samples", type=int, default=None)
    p.add_argument("--eta", type=float, default=None)
    p.add_argument("--v2", action="store_true", help="the DWT-Var / DCT-Var path of sample_condition_openai_v2.py")
    p.add_argument("--spatial-var", dest="spatial_var", action="store_true", help="(v2) pixel-space instead of transform-space variance")
    # MI355X build only
    p.add_argument("--dtype", choices=["bf16", "f32", "bf16x3", "f16x3"], default="f16x3",
                   help="UNet arithmetic: f16x3 (default) / bf16x3 = fp32 storage + split-precision convs (fp16- / bf16-headed; f16x3 polls a two-sided fp16-window watch per call and "
                        "redoes a flagged call in bf16x3), the reference's fp32 results to 1e-3 dB at 2.3x / 2.2x the speed of f32 (exact-f32 MFMA); bf16 = throughput mode, 2.3x faster again, |dPSNR| ~1e-2 dB")
    p.add_argument("--synthetic-weights", action="store_true", help="seeded random-init weights when the checkpoint is absent")
    p.add_argument("--synthetic-data", type=int, default=0, metavar="N", help="N seeded smooth images when the dataset folder is absent")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--lpips-checkpoint", type=str, default="", help="state_dict of lpips.LPIPS(net='vgg') (VGG-16 backbone + lin layers); "
                   "'synthetic' = seeded random weights (exercises the path, the value is meaningless); empty = no lpips key.  "
                   "Needs a power-of-two image size >= 128 (checked at start-up)")
    p.add_argument("--cpu-rng", action="store_true", help="draw the measurement noise and x_T from torch's CPU generator (the random stream of "
                   "the reference's CPU path): a run is then comparable value for value with a run of the reference on the same seeds")
    p.add_argument("--streams", type=int, default=1, help="split each sampler batch into this many part-batches, each with its own "
                   "UNet handle / operator context / HIP stream / host thread (overlaps HBM-bound and MFMA-bound phases)")
    p.add_argument("--x3-window", choices=["vjp", "launch"], default="vjp", help="split-precision modes: fp16 window of the VJP's gradient operands -- one scale per VJP (default) or one per "
                   "dgrad launch (networks whose gradients span more than one window, e.g. the ImageNet-256 architecture at high sigma: avoids f16x3's bf16-headed redo calls)")
    p.add_argument("--stream-offset-ms", type=float, default=10.0, help="with --streams > 1: part-batch k starts k x this many milliseconds after part 0, so that the "
                   "streams do not run the same phase of the UNet (large maps / small maps) at the same time; about a quarter of one guided call (bench.py measures it)")
    args = p.parse_args()

    if args.cpu_rng:
        km.set_noise_rng("cpu")
    config = load_json(args.config)
    model_config, dataset_config = config["model"], config["dataset"]
    v2 = args.v2 or "ortho_tf_type" in model_config
    if args.mle_sigma_thres is None:
        args.mle_sigma_thres = 1.0 if v2 else 0.2
    assert len(model_config["input_size"]) == 2 and model_config["input_size"][0] == model_config["input_size"][1]
    size = model_config["input_size"]

    env = ke.DistEnv()
    device = env.device
    if env.is_main_process:
        print("Using device:", device, flush=True)

    nstreams = max(1, args.streams)
    models = [ku.create_model_and_diffusion(image_size=size[0], dtype=args.dtype, device=device, **model_config["openai"]) for _ in range(nstreams)]
    inner_model, diffusion = models[0]
    if os.path.exists(args.checkpoint):
        sd = ku.normalize_state_dict(torch.load(args.checkpoint, map_location="cpu"))   # plain .pt or Lightning .ckpt layout
    elif args.synthetic_weights:
        sd = ku.synthetic_state_dict(seed=args.seed, out_cov=v2, image_size=size[0], model_channels=model_config["openai"]["num_channels"],
                                     num_res_blocks=model_config["openai"]["num_res_blocks"],
                                     attention_resolutions=model_config["openai"]["attention_resolutions"],
                                     **({"channel_mult": tuple(int(c) for c in model_config["openai"]["channel_mult"].split(","))}
                                        if model_config["openai"].get("channel_mult") else {}))
    else:
        raise FileNotFoundError(f"checkpoint {args.checkpoint} not found (pass --synthetic-weights for random-init weights)")
    for m, _ in models:
        m.load_state_dict(sd)
        if args.x3_window == "launch" and args.dtype in ("f16x3", "bf16x3"):
            m.set_x3_window("launch")
    sigma_min, sigma_max = model_config["sigma_min"], model_config["sigma_max"]

    if os.path.isdir(dataset_config["location"]):
        images = folder_of_images(dataset_config["location"])
    elif args.synthetic_data > 0:
        images = synthetic_images(args.synthetic_data, size[0])
    else:
        raise FileNotFoundError(f"dataset folder {dataset_config['location']} not found (pass --synthetic-data N)")

    operator_config = load_yaml(args.operator_config)
    import numpy as np
    np.random.seed(args.seed)                      # same mask on every rank and in every run (masks come from numpy's global stream)
    rng_state = np.random.get_state()
    trng_state = torch.get_rng_state()
    operators = []
    for _ in range(nstreams):                      # identical operators (same mask draw), one device context per stream
        np.random.set_state(rng_state)
        torch.set_rng_state(trng_state)
        operators.append(km.get_operator(device=device, **operator_config))
    operator = operators[0]
    if env.is_main_process:
        print(f"Operation: {operator_config['name']} / sigma_s: {operator_config['sigma_s']}")
        os.makedirs(args.logdir, exist_ok=True)
        save_yaml(vars(args), os.path.join(args.logdir, "args.yaml"))

    sigmas = ks.get_sigmas_karras(args.steps, sigma_min, sigma_max, rho=7.0, device="cpu")
    recon_mse = None
    if args.xstart_cov_type == "analytic":
        path = model_config.get("recon_mse", "")
        if os.path.exists(path):
            recon_mse = torch.load(path, map_location="cpu")
        else:   # synthetic table when the estimator (kdip_amd.analytic_variance) has not been run
            s = ks.get_sigmas_karras(1000, sigma_min, sigma_max, device="cpu")[:-1]
            recon_mse = {"sigmas": s, "mse_list": s ** 2 / (1 + s ** 2) * 0.5}

    loss_fn_vgg = None
    if args.lpips_checkpoint:
        import kdip_amd.lpips as klp
        klp.LPIPS.check_size(size[0], size[0])                 # fail at start-up, not after sampling, when the image size has no LPIPS path
        loss_fn_vgg = klp.LPIPS(net="vgg", device=device)
        loss_fn_vgg.load_state_dict(klp.synthetic_state_dict(args.seed) if args.lpips_checkpoint == "synthetic"
                                    else torch.load(args.lpips_checkpoint, map_location="cpu"))
    metrics_list = []
    for i, x0 in enumerate(images):
        x0 = x0[None].to(device)
        # the measurement (its noise) is a property of image i, identical on every rank: all ranks' samples are posterior
        # samples of the SAME measurement, the one rank 0 saves and scores; x_T and churn noise are then seeded per rank
        torch.manual_seed(args.seed * 1000003 + i)
        measurement = operator.forward(x0.clone(), flatten=True)
        torch.manual_seed(args.seed + 7919 * (i + 1) + env.rank)
        def make_model(k):
            if v2:
                from kdip_amd.external import OpenAIDenoiserV2
                denoiser = OpenAIDenoiserV2(models[k][0], diffusion, device=device, ortho_tf_type=model_config.get("ortho_tf_type"))
                return kc.ConditionOpenAIDenoiserV2(
                    denoiser=denoiser, operator=operators[k], measurement=measurement, guidance=args.guidance, device=device, zeta=args.zeta,
                    lambda_=args.lam, eta=args.eta, num_hutchinson_samples=args.num_hutchinson_samples, mle_sigma_thres=args.mle_sigma_thres,
                    ortho_tf_type=None if args.spatial_var else model_config.get("ortho_tf_type"))
            return kc.ConditionOpenAIDenoiser(
                inner_model=models[k][0], diffusion=diffusion, operator=operators[k], measurement=measurement, guidance=args.guidance,
                x0_cov_type=args.xstart_cov_type, recon_mse=recon_mse, lambda_=args.lam, zeta=args.zeta, eta=args.eta,
                num_hutchinson_samples=args.num_hutchinson_samples, mle_sigma_thres=args.mle_sigma_thres, device=device)

        cond_models = [make_model(k) for k in range(nstreams)]

        def sample_part(model, x):
            sampler = partial(ks.sample_heun if not args.euler else ks.sample_euler, model, x, sigmas, disable=True)
            if not args.ode:
                return sampler(s_churn=80, s_tmin=0.05, s_tmax=50, s_noise=1.003)
            return sampler()

        def sample_fn(n):
            if args.cpu_rng:
                x = torch.randn([n, model_config["input_channels"], size[0], size[1]]).to(device) * sigma_max
            else:
                x = torch.randn([n, model_config["input_channels"], size[0], size[1]], device=device) * sigma_max
            k = min(nstreams, n)
            if k == 1:
                return sample_part(cond_models[0], x)
            torch.cuda.synchronize()
            chunks = x.chunk(k)
            outs = ke.run_on_streams([lambda m=cond_models[j], c=chunks[j]: sample_part(m, c.contiguous()) for j in range(len(chunks))], device=device,
                                     delays=[j * args.stream_offset_ms * 1e-3 for j in range(len(chunks))] if args.stream_offset_ms > 0 else None)
            return torch.cat(outs)

        hat_x0 = ke.compute_features(env, sample_fn, lambda x: x, args.n, args.batch_size)
        metrics = kmet.compute_metrics(hat_x0, x0, loss_fn_vgg)
        metrics_list.append(metrics)
        if env.is_main_process:
            print(i, metrics, flush=True)
            if args.save_img:
                to_pil_image(measurement[0][0]).save(os.path.join(args.logdir, f"{args.prefix}_img_{i}_measurement.png"))
                for j, out in enumerate(hat_x0):
                    to_pil_image(out).save(os.path.join(args.logdir, f"{args.prefix}_img_{i}_hat_x0_sample_{j}.png"))

    avg_metrics = kmet.calculate_average_metric(metrics_list)
    if env.is_main_process:
        print(avg_metrics)
        save_yaml(avg_metrics, os.path.join(args.logdir, "avg_metrics.yaml"))
    env.barrier()


if __name__ == "__main__":
    main()
