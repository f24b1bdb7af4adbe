#!/usr/bin/env python3
"""Drop-in for the reference's `python fusion_generation/fusion_sampling.py ...` (Custom-Diffusion weights).

Accepts the reference's argv unchanged (flags of fusion_sampling.py:534-585; `+`-separated lists, background
concept last) and runs the same Tweedie-mix loop on the MI355X-native sampler.  What the reference does around
the loop through the HF hub (checkpoint download) takes local paths here:

  --sd_path              local SDXL checkpoint in diffusers layout (unet/, text_encoder/, text_encoder_2/, tokenizer/,
                         tokenizer_2/, optionally vae/): prompts are tokenised and encoded natively (tweediemix_amd/text.py,
                         modifier tokens of --personal_checkpoint injected), the UNet comes from unet/
  --vae_path             VAE folder or weights file (the reference uses madebyollin/sdxl-vae-fp16-fix); with a VAE the
                         final image is written as {output_path_all}/{prompt_orig}_{seed}.png like the reference
  --unet_path            diffusers-format SDXL UNet weights (.safetensors / torch state dict); the concept
                         checkpoints of --personal_checkpoint (delta-*.bin, key 'unet') load unchanged
  --text_embeds_path     precomputed embeddings instead of the text towers: torch file
                         {'text_embeds': (E[K+2,77,2048], P[K+2,1280]), 'text_embeds_single': (E[K,77,2048], P[K,1280])}
  --mask_paths           '+'-separated 8-bit masks for the foreground concepts (what run_expand.py would write
                         as '<concept>.jpg'); --random_masks draws seeded rectangles instead
  --synthetic            random-init weights / embeddings of the SDXL shapes (no checkpoints exist offline)
  --num_seeds N          N trajectories, seeds seed..seed+N-1 (trajectory i is exactly what `--seed seed+i` alone produces: its
                         x_T comes from its own generator), co-batched --seeds_per_batch at a time
  --gpus G               shard those seeds round-robin over G GPUs of this node: the script starts one process per GPU itself
                         (or runs under torch.distributed.run), every rank samples its seeds, the final latents are gathered
                         over RCCL and rank 0 writes the files (BASELINE config 4; the reference is single-GPU, sample_catdog.sh:3)

Output: {output_path_all}/{prompt_orig}_{seed}.latent.pt, plus the .png when VAE weights are given.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LORA = False


def build_parser():
    p = argparse.ArgumentParser()
    # --- the reference's flags, same names / defaults (fusion_sampling.py:534-585); output_path / output_path_all have no
    # default there (None crashes at os.makedirs), here they fall back to ./results{,_all}
    p.add_argument('--seed', type=int, default=182)
    p.add_argument('--device', type=str, default='cuda:0')
    p.add_argument('--output_path', type=str, default='results')
    p.add_argument('--output_path_all', type=str, default='results_all')
    p.add_argument('--negative_prompt', type=str, default='blurry, ugly, black, low res, unrealistic, blurry face')
    p.add_argument('--sd_version', type=str, default='2.1', choices=['1.4', '1.5', '2.0', '2.1', 'xl'])
    p.add_argument('--t_cond', type=float, default=0.4)
    p.add_argument('--guidance_scale', type=float, default=9.0)
    p.add_argument('--n_timesteps', type=int, default=50)
    p.add_argument('--prompt', type=str, default='')
    p.add_argument('--prompt_orig', type=str, default='')
    p.add_argument('--seg_concepts', type=str, default='')
    p.add_argument('--personal_checkpoint', type=str, default='')
    p.add_argument('--concepts', type=str, default='')
    p.add_argument('--modifier_token', type=str, default='')
    p.add_argument('--resampling_steps', type=int, default=10)
    p.add_argument('--jumping_steps', type=int, default=5)
    p.add_argument('--seg_gpu', type=int, default=1)
    p.add_argument('--crops_coords_top_left_h', type=int, default=0)
    p.add_argument('--crops_coords_top_left_w', type=int, default=0)
    p.add_argument('--resolution_h', type=int, default=1024)
    p.add_argument('--resolution_w', type=int, default=1024)
    if LORA:
        p.add_argument('--t_stop', type=float, default=0.9)          # fusion_sampling_lora.py:547
    # --- additive flags
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--sd_path', type=str, default='')
    p.add_argument('--vae_path', type=str, default='')
    p.add_argument('--unet_path', type=str, default='')
    p.add_argument('--text_embeds_path', type=str, default='')
    p.add_argument('--mask_paths', type=str, default='')
    p.add_argument('--random_masks', action='store_true')
    p.add_argument('--num_seeds', type=int, default=1, help='trajectories to sample: seeds seed..seed+n-1')
    p.add_argument('--seeds_per_batch', type=int, default=0, help='seeds co-batched into every UNet launch (0: all of this rank\'s seeds, at most 4)')
    p.add_argument('--gpus', type=int, default=1, help='shard the seeds over this many GPUs (one process per GPU, started by this script)')
    p.add_argument('--no_strict_reference', action='store_true',
                   help='route the concept weights for any concept count (the reference hooks only route when the UNet batch is 4, i.e. 3 concepts)')
    p.add_argument('--streams', type=int, default=1, help='launch chains per UNet call (2: batch rows split over two HIP streams)')
    p.add_argument('--lora_mode', type=str, default='merged', choices=['merged', 'lowrank'],
                   help='LoRA deltas as merged per-concept weight sets (default, fastest) or in the reference\'s own low-rank form up(down(x)) (no weight copies)')
    p.add_argument('--dtype', type=str, default='bf16', choices=['bf16', 'fp8'],
                   help='fp8: the transformer blocks\' projections, the attention outputs and the eligible ResnetBlock2D convolutions on e4m3 operands with power-of-two block scales (merged LoRA only)')
    p.add_argument('--no_graphs', action='store_true')
    p.add_argument('--tiny', action='store_true', help='tiny UNet config (smoke tests)')
    return p


def load_state_dict(path):
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        return load_file(path)
    sd = torch.load(path, map_location='cpu')
    return sd.get('state_dict', sd)


def find_weights(folder, stem):
    """first existing of {stem}.fp16.safetensors / {stem}.safetensors / {stem}.bin under folder (or folder itself if a file)."""
    if os.path.isfile(folder):
        return folder
    for ext in ('.fp16.safetensors', '.safetensors', '.bin'):
        fp = os.path.join(folder, stem + ext)
        if os.path.exists(fp):
            return fp
    raise FileNotFoundError(f'no {stem}.[fp16.]safetensors/.bin under {folder}')


def save_png(img, path):
    """img [3,H,W] in [0,1] -> 8-bit PNG (image_processor.postprocess(..., 'pil') of fusion_sampling.py:526-527)."""
    from PIL import Image
    a = (img.clamp(0, 1).permute(1, 2, 0).float().cpu().numpy() * 255).round().astype('uint8')
    Image.fromarray(a).save(path)


def noise_for_seed(seed, h, w):
    """x_T of one trajectory, drawn on the CPU like fusion_sampling.py:488 after seed_everything(seed) (utils_custom.py:10-14
    seeds torch's global generator; a fresh generator with the same seed yields the same first draw)."""
    return torch.randn(1, 4, h, w, generator=torch.Generator().manual_seed(int(seed)))


def main(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.dtype == 'fp8' and opt.lora_mode == 'lowrank':
        raise SystemExit('--dtype fp8 quantises the merged per-concept projection weights: use --lora_mode merged')
    from tweediemix_amd import dist as D, launch as LA, masks as M, sampler as S, unet as U, weights as Wt
    if opt.gpus > 1 and not LA.launched():
        return LA.self_launch(opt.gpus)
    rank, local, world = LA.rank_env()
    if world > 1:
        single = bool(os.environ.get('TMIX_SINGLE_GPU_DIST_TEST'))      # tests: all ranks on GPU 0, gloo
        opt.device = 'cuda:0' if single else f'cuda:{local}'
        torch.cuda.set_device(torch.device(opt.device))
        # through dist.init like bench.py: a rendezvous port stolen between the launcher's probe and the bind exits with EADDRINUSE_RC,
        # which launch.self_launch retries on a fresh port
        D.init(torch.device(opt.device), world, backend='gloo' if single else None)
    say = print if rank == 0 else (lambda *a, **k: None)
    if opt.sd_version != 'xl':
        say(f"note: --sd_version {opt.sd_version}: like the reference (fusion_sampling.py:119) only the SDXL pipeline exists")
    concepts = [c for c in opt.concepts.split('+') if c] or ['a', 'b', 'background']
    K = len(concepts)                                    # concept_num, background last (fusion_sampling.py:143-148)
    cfg = U.TINY if opt.tiny else U.SDXL
    if opt.sd_path and os.path.exists(os.path.join(opt.sd_path, 'unet', 'config.json')):
        import json
        cfg = U.UNetConfig.from_diffusers(json.load(open(os.path.join(opt.sd_path, 'unet', 'config.json'))))
    kind = 'lora' if LORA else 'custom'
    S.seed_everything(opt.seed)
    if opt.synthetic:
        sd = Wt.synthetic_state_dict(cfg, seed=1234, device=opt.device, dtype=torch.bfloat16)
        con = Wt.synthetic_concepts(cfg, kind, K, device=opt.device)
        g = torch.Generator().manual_seed(42)
        te = (torch.randn(K + 2, 77, cfg.cross_dim, generator=g), torch.randn(K + 2, cfg.pooled_dim, generator=g))
        ts = (torch.randn(K, 77, cfg.cross_dim, generator=g), torch.randn(K, cfg.pooled_dim, generator=g))
    else:
        unet_file = opt.unet_path or (opt.sd_path and find_weights(os.path.join(opt.sd_path, 'unet'), 'diffusion_pytorch_model'))
        if not (unet_file and (opt.text_embeds_path or opt.sd_path) and opt.personal_checkpoint):
            sys.exit("need --personal_checkpoint and either --sd_path (local diffusers-layout SDXL checkpoint) or "
                     "--unet_path + --text_embeds_path (or --synthetic); there is no HF hub download here")
        sd = load_state_dict(unet_file)
        sts = [torch.load(p, map_location='cpu') for p in opt.personal_checkpoint.split('+')]            # :156-157
        con = [st['unet'] for st in sts]
        if opt.text_embeds_path:
            emb = torch.load(opt.text_embeds_path, map_location='cpu')
            te, ts = emb['text_embeds'], emb['text_embeds_single']
        else:
            from tweediemix_amd import text as T
            te, ts, K_text = T.TextPath(opt.sd_path, opt.device).embed(opt, sts)
            assert K_text == K, (K_text, K)
    W = U.UNetWeights(cfg, sd, opt.device, (kind, con), lora_mode=opt.lora_mode)
    h, w = opt.resolution_h // 8, opt.resolution_w // 8
    sidecar = False
    if opt.mask_paths:
        fg = opt.mask_paths.split('+')
    elif opt.random_masks or opt.synthetic:
        fg = None                                        # seeded rectangles, one set per trajectory seed (drawn when the sampler asks)
    else:   # the reference's file contract (:453-466): the side-car writes '<seg_concept>.jpg' under output_path
        # (several ranks: one side-car directory per rank, see masks.sidecar_layout)
        side_dir, side_gpu = M.sidecar_layout(opt.output_path, rank, world, local, opt.seg_gpu)
        fg = [os.path.join(side_dir, sp + '.jpg') for sp in opt.seg_concepts.split('+')]
        sidecar = True
    vae, vae_scaling = None, None
    vae_dir = opt.vae_path or (opt.sd_path and os.path.isdir(os.path.join(opt.sd_path, 'vae')) and os.path.join(opt.sd_path, 'vae'))
    if vae_dir:
        from tweediemix_amd import vae as V
        vcfg = V.FULL
        if os.path.isdir(vae_dir) and os.path.exists(os.path.join(vae_dir, 'config.json')):
            import json
            j = json.load(open(os.path.join(vae_dir, 'config.json')))
            vcfg = dict(block_out_channels=tuple(j['block_out_channels']), layers_per_block=j.get('layers_per_block', 2),
                        latent_channels=j.get('latent_channels', 4), out_channels=j.get('out_channels', 3),
                        groups=j.get('norm_num_groups', 32))
            vae_scaling = j.get('scaling_factor')
        vae = (vcfg, load_state_dict(find_weights(vae_dir, 'diffusion_pytorch_model')))
    elif opt.synthetic and opt.tiny:
        from tweediemix_amd import vae as V
        vae = (V.TINY, V.synthetic_state_dict(V.TINY))
    seeds = D.seed_shard([opt.seed + i for i in range(opt.num_seeds)], rank, world)
    per = opt.seeds_per_batch or min(max(len(seeds), 1), 4)
    strict = not opt.no_strict_reference
    if strict and K + 1 != 4 and kind in ('custom', 'lora'):
        say(f"note: {K} concepts -> UNet batch {K + 1}: the reference's attention hooks only route concept weights when the batch is 4 "
            f"(utils_custom.py:62, utils_lora.py:63), so like the reference this run uses the BASE weights for every row; "
            f"--no_strict_reference routes them")
    current = {"ids": [opt.seed], "turn": 0}             # the seeds of the batch being sampled; the sampler asks once per seed, in order

    def provider(x0):
        if fg is not None:
            return M.build_masks(fg, h, w, opt.device)
        sd_ = current["ids"][current["turn"] % len(current["ids"])]
        current["turn"] += 1
        return M.build_masks(M.random_rectangle_masks(K, opt.resolution_h, opt.resolution_w, seed=sd_), h, w, opt.device)

    tw = S.Tweediemix(opt, W, te, ts, provider, concept_num=K, lora=LORA,
                      strict_reference=strict, use_graphs=not opt.no_graphs, n_seeds=per, n_streams=opt.streams, vae=vae,
                      fp8=(opt.dtype == 'fp8'))
    if vae_scaling:                                       # fusion_sampling.py:518 divides by vae.config.scaling_factor
        tw.vae_scaling_factor = float(vae_scaling)
    if sidecar and vae is not None:
        # with a VAE the whole contract runs like the reference: decode the Tweedie preview to {output_path}/tweedie.jpg,
        # call `CUDA_VISIBLE_DEVICES={seg_gpu} python text_segment/run_expand.py ...` (TMIX_SEG_CMD overrides the command),
        # read the masks back; without one the mask files must already be there
        tw.mask_provider = M.SidecarMaskProvider(tw, side_dir, opt.seg_concepts, seg_gpu=side_gpu,
                                                 cmd_template=os.environ.get('TMIX_SEG_CMD'))
    lats, imgs = [], []
    for b0 in range(0, len(seeds), per):
        batch = seeds[b0:b0 + per]
        ids = (batch + batch * per)[:per]                # a ragged last batch is padded with repeats and trimmed below
        current["ids"], current["turn"] = ids, 0
        lat_b = tw.run_fusion(torch.cat([noise_for_seed(sd_, h, w) for sd_ in ids]))
        lats.append(lat_b[:len(batch)])
        if vae is not None:                               # fusion_sampling.py:496-528
            imgs.append(tw.decode_final(lat_b)[:len(batch)])
    dev = torch.device(opt.device)
    lat = torch.cat(lats) if lats else torch.zeros(0, 4, h, w, device=dev)
    img = torch.cat(imgs) if imgs else None
    if world > 1:                                         # the result gather: the only collective of this path
        import torch.distributed as dist
        lat = D.gather_latents(lat.contiguous(), opt.num_seeds, rank, world)
        if vae is not None:
            img = D.gather_latents(img.contiguous() if img is not None else torch.zeros(0, 3, opt.resolution_h, opt.resolution_w, device=dev),
                                   opt.num_seeds, rank, world)
    if rank == 0:
        os.makedirs(opt.output_path_all, exist_ok=True)
        prompt_orig = opt.prompt_orig.split('+')[0] or 'sample'
        for i in range(opt.num_seeds):
            out = f'{opt.output_path_all}/{prompt_orig}_{opt.seed + i}.latent.pt'
            torch.save(lat[i:i + 1].cpu(), out)
            print('saved', out)
            if vae is not None:
                out = f'{opt.output_path_all}/{prompt_orig}_{opt.seed + i}.png'
                save_png(img[i], out)
                print('saved', out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return lat


if __name__ == '__main__':
    rc = main()
    sys.exit(rc if isinstance(rc, int) else 0)
