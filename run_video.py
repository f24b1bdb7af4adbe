#!/usr/bin/env python3
"""Drop-in for the reference's run_video.py (I2VGen-XL image-to-video from the fused image, BASELINE config #5).

The reference script hard-codes its inputs and pulls `ali-vilab/i2vgen-xl` from the hub; here the same settings are flags
with the reference's values as defaults, and the model comes from local files:

  --i2v_path           diffusers-layout I2VGen-XL folder: unet/, tokenizer/ + text_encoder/, image_encoder/ (+ feature_extractor/),
                       vae/ -- prompt embeddings, the CLIP image embedding and the image latents are computed natively from
                       --prompt / --negative_prompt / --image_path (tweediemix_amd/text.py, vae.py, video.py)
  --conditioning_path  optional torch file with any of {'prompt_embeds': [2,77,1024] (negative row first), 'image_embeddings':
                       [2,1024] (zeros row first), 'image_latents': [2,4,F,h,w]} to use instead of computing them
  --alphas_cumprod     .npy with the checkpoint scheduler's 1000-entry table (else: cosine schedule with zero terminal SNR)
  --synthetic          random-init network and conditioning of the real shapes (no checkpoints exist offline)

Per step: native UNet forward on the CFG pair (tweediemix_amd/i2vgen.py) with the first-frame feature injection of
video_gen/utils_attn.py:389-474 for the first int(steps * injection_timestep) steps, then the fused CFG / v-prediction /
DDIM update (tmix_vpred_step).  Output: output_i2v_seed_<seed>.latent.pt, plus output_i2v_seed_<seed>.gif with --vae_path."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser()
    # the reference script's constants (run_video.py:7-38)
    p.add_argument("--image_path", default="test_out/photo of a cat and a dog running, mountain background_3821.png")
    p.add_argument("--prompt", default="A cat and a dog running, mountain background")
    p.add_argument("--negative_prompt", default="Distorted, discontinuous, Ugly, blurry, low resolution, motionless, static, disfigured, "
                                                "disconnected limbs, Ugly faces, incomplete arms")
    p.add_argument("--seed", type=int, default=6425)
    p.add_argument("--num_inference_steps", type=int, default=50)
    p.add_argument("--guidance_scale", type=float, default=9.0)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--target_fps", type=int, default=8)
    p.add_argument("--num_frames", type=int, default=16)
    p.add_argument("--injection_timestep", type=float, default=0.02)
    p.add_argument("--interp_ratio", type=float, default=0.7)
    # local inputs
    p.add_argument("--i2v_path", default="")
    p.add_argument("--conditioning_path", default="")
    p.add_argument("--alphas_cumprod", default="")
    p.add_argument("--vae_path", default="")
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--tiny", action="store_true", help="tiny network (smoke tests)")
    p.add_argument("--no_graphs", action="store_true")
    p.add_argument("--streams", type=int, default=2, choices=[1, 2], help="2: the two clips of the CFG pair run as two launch chains")
    return p


def native_conditioning(opt, gen, have):
    """what I2VGenXLPipeline.__call__ computes before its loop (video_gen/pipeline_i2vgen_xl.py:604-639), from the checkpoint's
    tokenizer/ + text_encoder/ (prompt embeddings), image_encoder/ (CLIP image embedding) and vae/ (image latents); entries
    already present in the --conditioning_path file are kept."""
    import json
    from PIL import Image
    from fusion_generation.fusion_sampling import find_weights, load_state_dict
    from tweediemix_amd import text as T, vae as VA, video as V
    out = {}
    root = opt.i2v_path
    if "prompt_embeds" not in have:                     # encode_prompt: negative row first (CFG order)
        tok = T.ClipBPETokenizer.from_pretrained(os.path.join(root, "tokenizer"))
        enc = T.load_text_tower(os.path.join(root, "text_encoder"))
        out["prompt_embeds"] = enc.last_hidden_state(tok([opt.negative_prompt, opt.prompt])).float().cpu()
    if "image_embeddings" in have and "image_latents" in have:
        return out
    image = Image.open(opt.image_path).convert("RGB")
    if "image_embeddings" not in have:                  # :623-628 + _encode_image: crop to (w, w), bilinear to the tower's size, CLIP stats
        icfg = json.load(open(os.path.join(root, "image_encoder", "config.json")))
        fx = os.path.join(root, "feature_extractor", "preprocessor_config.json")
        fcfg = json.load(open(fx)) if os.path.exists(fx) else {}
        size = icfg.get("image_size", 224)
        tower = T.ClipVisionEncoder(load_state_dict(find_weights(os.path.join(root, "image_encoder"), "model")), icfg["num_attention_heads"],
                                    icfg["patch_size"], icfg.get("hidden_act", "gelu"), icfg.get("layer_norm_eps", 1e-5))
        crop = V.resize_bilinear(V.center_crop_wide(image, (opt.width, opt.width)), (size, size))
        kw = {k: tuple(fcfg[k2]) for k, k2 in (("mean", "image_mean"), ("std", "image_std")) if k2 in fcfg}
        emb = tower(V.clip_pixel_values(crop, **kw)).float().cpu()
        out["image_embeddings"] = torch.cat([torch.zeros_like(emb), emb])
    if "image_latents" not in have:                     # :631-639 prepare_image_latents
        vdir = os.path.join(root, "vae")
        j = json.load(open(os.path.join(vdir, "config.json")))
        vcfg = dict(block_out_channels=tuple(j["block_out_channels"]), layers_per_block=j.get("layers_per_block", 2),
                    latent_channels=j.get("latent_channels", 4), out_channels=j.get("out_channels", 3), groups=j.get("norm_num_groups", 32))
        encp = VA.VAEEncoderPlan(vcfg, load_state_dict(find_weights(vdir, "diffusion_pytorch_model")), 1, opt.height, opt.width)
        mean, logvar = encp(V.vae_pixel_values(V.center_crop_wide(image, (opt.width, opt.height))).cuda())
        sample = mean.cpu() + torch.exp(0.5 * logvar.cpu()) * torch.randn(mean.shape, generator=gen)       # latent_dist.sample()
        out["image_latents"] = V.prepare_image_latents(sample, opt.num_frames, j.get("scaling_factor", 0.18215))
    return out


def main(argv=None):
    opt = build_parser().parse_args(argv)
    from tweediemix_amd import i2vgen as I, video as V
    cfg = I.TINY if opt.tiny else I.FULL
    h, w, Fr = opt.height // 8, opt.width // 8, opt.num_frames
    if Fr != 16:
        print("note: the reference's injection hook hard-codes 16 frames (video_gen/utils_attn.py:439)")
    gen = torch.Generator().manual_seed(opt.seed)
    if opt.synthetic:
        from tweediemix_amd.weights import synthetic_i2vgen_state_dict
        sd = synthetic_i2vgen_state_dict(cfg)
        cond = {"prompt_embeds": torch.randn(2, 77, cfg.cross_dim, generator=gen), "image_embeddings": torch.randn(2, cfg.cross_dim, generator=gen),
                "image_latents": torch.randn(2, 4, Fr, h, w, generator=gen)}
    else:
        if not opt.i2v_path:
            sys.exit("need --i2v_path (or --synthetic); there is no hub download here")
        from fusion_generation.fusion_sampling import find_weights, load_state_dict
        sd = load_state_dict(find_weights(os.path.join(opt.i2v_path, "unet"), "diffusion_pytorch_model"))
        cond = torch.load(opt.conditioning_path, map_location="cpu") if opt.conditioning_path else {}
        cond.update(native_conditioning(opt, gen, cond))
    Wt = I.I2VWeights(cfg, sd)
    fps = torch.tensor([float(opt.target_fps)] * 2)
    fe, ctx, ilf = I.conditioning(Wt, fps, cond["image_latents"], cond["image_embeddings"], cond["prompt_embeds"])
    plan = (I.I2VPlanGroup if opt.streams == 2 else I.I2VPlan)(Wt, 2, Fr, h, w, fe, ctx, ilf, interp=opt.interp_ratio)
    sched_json = os.path.join(opt.i2v_path, "scheduler", "scheduler_config.json") if opt.i2v_path else ""
    sch_kw = {}
    if opt.alphas_cumprod:
        acp = np.load(opt.alphas_cumprod).astype(np.float32)
    elif sched_json and os.path.exists(sched_json):      # the checkpoint's own DDIMScheduler settings (pipeline_i2vgen_xl.py:480)
        import json
        acp, sch_kw = V.alphas_from_scheduler_config(json.load(open(sched_json)))
    else:                                   # stand-in: the published i2vgen-xl scheduler settings (squaredcos_cap_v2, zero terminal SNR)
        if not opt.synthetic:
            print("warning: no scheduler/scheduler_config.json under --i2v_path and no --alphas_cumprod: using the published "
                  "i2vgen-xl settings (squaredcos_cap_v2, rescale_betas_zero_snr, steps_offset 1, set_alpha_to_one False)")
        acp, sch_kw = V.alphas_from_scheduler_config(dict(beta_schedule="squaredcos_cap_v2", rescale_betas_zero_snr=True,
                                                          steps_offset=1, set_alpha_to_one=False))
    sch = V.VideoSchedule(acp, opt.num_inference_steps, **sch_kw)
    inj = V.FeatureInjector(sch.injection_schedule(opt.injection_timestep), opt.interp_ratio, clips=2, frames=Fr)
    x = torch.randn(1, 4, Fr, h, w, generator=gen).cuda()          # latents * init_noise_sigma (= 1 for DDIM)
    graphs = {}

    def unet(xin, t):                                               # one recorded forward per injection state, replayed as a hipGraph
        if opt.streams == 2:
            plan.set_input(xin, t)
        else:
            xv = plan.x_in.view(2, Fr, 2 * cfg.in_channels, h, w)
            xv[:, :, :cfg.in_channels] = xin.permute(0, 2, 1, 3, 4)
            plan.t_dev.fill_(float(t))
        if opt.no_graphs:
            plan.run()
        else:
            g = graphs.get(plan.inject)
            if g is None:
                plan.run(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    plan.run()
                graphs[plan.inject] = g
            g.replay()
        return plan.eps.view(2, Fr, cfg.out_channels, h, w).permute(0, 2, 1, 3, 4).contiguous()

    unet.plan = plan
    lat = V.sample_loop(unet, x, sch, opt.guidance_scale, inj)
    out = f"output_i2v_seed_{opt.seed}.latent.pt"
    torch.save(lat.cpu(), out)
    print("saved", out)
    if opt.vae_path:
        import json
        from PIL import Image
        from fusion_generation.fusion_sampling import find_weights, load_state_dict
        from tweediemix_amd import vae as VA
        vcfg = VA.FULL
        if os.path.isdir(opt.vae_path) and os.path.exists(os.path.join(opt.vae_path, "config.json")):
            j = json.load(open(os.path.join(opt.vae_path, "config.json")))
            vcfg = dict(block_out_channels=tuple(j["block_out_channels"]), layers_per_block=j.get("layers_per_block", 2),
                        latent_channels=j.get("latent_channels", 4), out_channels=j.get("out_channels", 3), groups=j.get("norm_num_groups", 32))
        dec = VA.VAEDecoderPlan(vcfg, load_state_dict(find_weights(opt.vae_path, "diffusion_pytorch_model")), 1, h, w, 1 / 0.18215, "cuda")
        frames = []
        for f in range(Fr):                                         # pipeline decode_latents: 1/scaling_factor, per frame
            img = dec(lat[:, :, f].contiguous())[0].clamp(0, 1)
            frames.append(Image.fromarray((img.permute(1, 2, 0).float().cpu().numpy() * 255).round().astype("uint8")))
        gif = f"output_i2v_seed_{opt.seed}.gif"
        frames[0].save(gif, save_all=True, append_images=frames[1:], duration=1000 // opt.target_fps, loop=0)
        print("saved", gif)
    return lat


if __name__ == "__main__":
    main()
