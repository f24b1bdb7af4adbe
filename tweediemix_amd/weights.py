"""Weight preparation for the HIP kernels (layout transforms done once at load time)."""
from __future__ import annotations

import torch


def interleave_geglu(w: torch.Tensor, b: torch.Tensor | None):
    """diffusers GEGLU: proj = Linear(C, 8C); value = out[:, :4C], gate = out[:, 4C:].
    The GEMM epilogue fuses value*gelu(gate) when the weight rows alternate in groups of 16:
    [value 0..15 | gate 0..15 | value 16..31 | gate 16..31 | ...]."""
    n2 = w.shape[0]
    n = n2 // 2
    assert n % 16 == 0
    idx = torch.arange(n2, device=w.device).reshape(2, n // 16, 16).permute(1, 0, 2).reshape(-1)
    return w[idx].contiguous(), (None if b is None else b[idx].contiguous())


def fold_layernorm(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, bias: torch.Tensor | None):
    """Linear(LayerNorm(x)) with the norm's affine folded into the Linear (tmix_gemm_desc.ln_*):
    returns (W' = W*gamma as bf16, colsum[n] = sum_k W'[n][k] of the bf16-rounded W', t[n] = W[n]·beta + bias[n]).
    w may be [N,K] or a stack [R,N,K] of per-row weight sets."""
    w32 = w.float()
    wp = (w32 * gamma.float()).to(torch.bfloat16)
    colsum = wp.float().sum(-1)
    t = w32 @ beta.float()
    if bias is not None:
        t = t + bias.float()
    return wp.contiguous(), colsum.contiguous(), t.contiguous()


# ----------------------------------------------------------------------------------------------
# parameter inventory (diffusers key scheme) and synthetic weights / concept deltas
# ----------------------------------------------------------------------------------------------
def param_shapes(cfg) -> dict:
    """name -> shape of every UNet2DConditionModel parameter for `cfg` (SDXL-base layout)."""
    P = {}
    C0, T = cfg.block_out_channels[0], cfg.time_embed_dim

    def lin(name, i, o, bias=True):
        P[name + ".weight"] = (o, i)
        if bias:
            P[name + ".bias"] = (o,)

    def conv(name, i, o, k=3):
        P[name + ".weight"] = (o, i, k, k)
        P[name + ".bias"] = (o,)

    def norm(name, c):
        P[name + ".weight"] = (c,)
        P[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", ci, co)
        lin(name + ".time_emb_proj", T, co)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co)
        if ci != co:
            conv(name + ".conv_shortcut", ci, co, 1)

    def t2d(name, c, n):
        norm(name + ".norm", c)
        lin(name + ".proj_in", c, c)
        for i in range(n):
            b = f"{name}.transformer_blocks.{i}"
            norm(b + ".norm1", c)
            for a, kd in (("attn1", c), ("attn2", cfg.cross_dim)):
                lin(f"{b}.{a}.to_q", c, c, False)
                lin(f"{b}.{a}.to_k", kd, c, False)
                lin(f"{b}.{a}.to_v", kd, c, False)
                lin(f"{b}.{a}.to_out.0", c, c)
            norm(b + ".norm2", c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
        lin(name + ".proj_out", c, c)

    conv("conv_in", cfg.in_channels, C0)
    lin("time_embedding.linear_1", C0, T)
    lin("time_embedding.linear_2", T, T)
    lin("add_embedding.linear_1", cfg.add_in_dim, T)
    lin("add_embedding.linear_2", T, T)
    ch = cfg.block_out_channels
    nb = len(ch)
    skip, ci = [C0], C0
    for bi, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{bi}.resnets.{j}", ci, co)
            if cfg.transformer_layers[bi]:
                t2d(f"down_blocks.{bi}.attentions.{j}", co, cfg.transformer_layers[bi])
            ci = co
            skip.append(co)
        if bi < nb - 1:
            conv(f"down_blocks.{bi}.downsamplers.0.conv", co, co)
            skip.append(co)
    cm = ch[-1]
    resnet("mid_block.resnets.0", cm, cm)
    t2d("mid_block.attentions.0", cm, cfg.transformer_layers[-1])
    resnet("mid_block.resnets.1", cm, cm)
    for ui, co in enumerate(reversed(ch)):
        bi = nb - 1 - ui
        for j in range(cfg.layers_per_block + 1):
            resnet(f"up_blocks.{ui}.resnets.{j}", ci + skip.pop(), co)
            if cfg.transformer_layers[bi]:
                t2d(f"up_blocks.{ui}.attentions.{j}", co, cfg.transformer_layers[bi])
            ci = co
        if ui < nb - 1:
            conv(f"up_blocks.{ui}.upsamplers.0.conv", co, co)
    norm("conv_norm_out", C0)
    conv("conv_out", C0, cfg.out_channels)
    return P


def synthetic_state_dict(cfg, seed=1234, device="cpu", nontrivial=False, dtype=torch.float32, hostile=False):
    """Random-init weights of the SDXL architecture (there are no checkpoints offline):
    Linear/conv ~ N(0, 1/fan_in), biases 0, norms (1, 0); residual-branch outputs (to_out, ff.net.2,
    conv2, proj_out) x0.1 so activations stay O(1) through 70 blocks.  Values are rounded to bf16
    (what the kernels consume) and returned in `dtype`.  nontrivial=True also randomises biases and
    norm affine parameters so tests exercise them.
    hostile=True (implies nontrivial): the activation statistics real checkpoints are known for and N(0, 1/fan_in) never shows -- see
    `make_hostile`."""
    nontrivial = nontrivial or hostile
    gen = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        is_norm = ".norm" in name or name.startswith("conv_norm_out")
        if name.endswith(".bias"):
            v = torch.randn(shape, generator=gen, device=device) * 0.1 if nontrivial else torch.zeros(shape, device=device)
        elif is_norm:
            v = torch.ones(shape, device=device)
            if nontrivial:
                v = v + 0.1 * torch.randn(shape, generator=gen, device=device)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = torch.randn(shape, generator=gen, device=device) * fan_in ** -0.5
            if any(k in name for k in (".to_out.0.", ".ff.net.2.", ".conv2.", ".proj_out.")):
                v = v * 0.1
        sd[name] = v.to(torch.bfloat16).to(dtype)
    if hostile:
        make_hostile(sd, seed)
    return sd


# what `hostile` plants, and where it lands in the kernels (VERDICT r3 "hostile statistics"):
HOSTILE = dict(row_mean=4.0,          # every channel of the residual stream of every other Transformer2DModel carries this offset: LayerNorm rows with
                                      # |mean| / std ~ 4 (the single-pass var = E[x^2] - mean^2 of the fused LayerNorm; a bf16 stream cannot carry much
                                      # more, see DESIGN "hostile statistics")
               outlier_frac=0.01,     # in the others this share of the channels ...
               outlier=150.0,         # ... sits at +-150 (x std ~ 1): "massive activations" -- they dominate row variance, the MX block scales of the e4m3
                                      # copies (a block of 32 holding one loses 8 binades for the other 31) and the GroupNorm groups they fall into
               group_mean=6.0,        # half of the GroupNorm groups of every conv output carry this offset (group mean / std ~ 6)
               gate_tail=-4.5,        # a quarter of the GEGLU gates sit here: gelu(-4.5) ~ -1.5e-5, the far tail of the erf form
               gate_frac=0.25)


def make_hostile(sd, seed=1234, groups=32):
    """in place: biases that give a synthetic checkpoint hostile activation statistics (HOSTILE).  Weights stay N(0, 1/fan_in), so every
    tensor is still O(1) in std; only the offsets change."""
    H = HOSTILE
    gen = torch.Generator(device="cpu").manual_seed(seed + 99)

    def outliers(n):
        k = max(1, int(round(n * H["outlier_frac"])))
        idx = torch.randperm(n, generator=gen)[:k]
        sgn = (torch.randint(0, 2, (k,), generator=gen) * 2 - 1).float()
        return idx, sgn

    def put(name, v):
        sd[name] = v.to(torch.bfloat16).to(sd[name].dtype).to(sd[name].device)

    n_t2d = 0
    for name in sorted(sd):
        if not name.endswith(".bias"):
            continue
        b = sd[name].float().cpu().clone()
        n = b.numel()
        if name.endswith("proj_in.bias"):                      # the residual stream of a Transformer2DModel starts here
            # (alternating: outlier channels dominate a row's variance, so a stream that carries them has |mean| / std < 1 whatever its offset)
            n_t2d += 1
            if n_t2d & 1:
                b += H["row_mean"]
            else:
                idx, sgn = outliers(n)
                b[idx] = sgn * H["outlier"]
        elif ".ff.net.0.proj.bias" in name:                    # diffusers GEGLU: hidden, gate = proj(x).chunk(2)
            half = n // 2
            k = int(half * H["gate_frac"])
            idx = torch.randperm(half, generator=gen)[:k]
            b[half + idx] = H["gate_tail"]
        elif ".to_out.0.bias" in name or ".ff.net.2.bias" in name:   # every residual branch pushes a few channels a little further
            idx, sgn = outliers(n)
            b[idx] += sgn * 2.0
        elif any(k in name for k in (".conv1.bias", ".conv2.bias", "conv_in.bias", ".downsamplers.0.conv.bias", ".upsamplers.0.conv.bias", ".proj_out.bias")) and n % groups == 0:
            cpg = n // groups
            gsel = torch.randperm(groups, generator=gen)[:groups // 2]
            for g_ in gsel.tolist():
                b[g_ * cpg:(g_ + 1) * cpg] += H["group_mean"]
            idx, sgn = outliers(n)
            b[idx] += sgn * (H["outlier"] if ".conv1." in name or name.startswith("conv_in") else 20.0)
        else:
            continue
        put(name, b)
    return sd


def attention_block_prefixes(cfg):
    return sorted({k.rsplit(".attn1", 1)[0] for k in param_shapes(cfg) if ".attn1.to_q" in k})


def synthetic_concepts(cfg, kind, K, device="cpu", dtype=torch.float32):
    """K concept checkpoints in the reference's own 'unet' key scheme:
    custom: '<tb>.attn2.to_k.weight' / 'to_v.weight' ([C,cross], seed 2000+i)   (diffusers_training_xl_new.py:45-66)
    lora  : '<tb>.attn{1,2}.processor.to_{q,k,v,out}_lora.{down,up}.weight', rank 4, down ~ N(0,1/4),
            up ~ N(0,0.02) (seed 3000+i)                                         (model_lora.py:28-48,104-115)"""
    shapes = param_shapes(cfg)
    out = []
    for i in range(K):
        sd = {}
        if kind == "custom":
            gen = torch.Generator(device=device).manual_seed(2000 + i)
            for tb in attention_block_prefixes(cfg):
                for nm in ("to_k", "to_v"):
                    shp = shapes[f"{tb}.attn2.{nm}.weight"]
                    sd[f"{tb}.attn2.{nm}.weight"] = (torch.randn(shp, generator=gen, device=device) * shp[1] ** -0.5).to(torch.bfloat16).to(dtype)
        elif kind == "lora":
            gen = torch.Generator(device=device).manual_seed(3000 + i)
            for tb in attention_block_prefixes(cfg):
                for a in ("attn1", "attn2"):
                    for nm, wn in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("out", "to_out.0")):
                        o, ii = shapes[f"{tb}.{a}.{wn}.weight"]
                        sd[f"{tb}.{a}.processor.to_{nm}_lora.down.weight"] = (torch.randn(4, ii, generator=gen, device=device) * 0.25).to(dtype)
                        sd[f"{tb}.{a}.processor.to_{nm}_lora.up.weight"] = (torch.randn(o, 4, generator=gen, device=device) * 0.02).to(dtype)
        else:
            raise ValueError(kind)
        out.append(sd)
    return out


def synthetic_clip_state_dict(d: int, layers: int, inter: int, vocab: int = 49408, max_pos: int = 77, proj: int | None = None,
                              seed: int = 77, dtype=torch.float32):
    """random-init text tower in transformers' CLIPTextModel key scheme (no checkpoints exist offline):
    Linear weights N(0, 1/fan_in), embeddings N(0, 0.02), norms gamma 1 + small noise / beta small noise."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(dtype)
    sd = {"text_model.embeddings.token_embedding.weight": rn(vocab, d, std=0.02),
          "text_model.embeddings.position_embedding.weight": rn(max_pos, d, std=0.01)}
    for i in range(layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{n}.weight"] = rn(d, d, std=d ** -0.5)
            sd[p + f"self_attn.{n}.bias"] = rn(d, std=0.02)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rn(inter, d, std=d ** -0.5), rn(inter, std=0.02)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rn(d, inter, std=inter ** -0.5), rn(d, std=0.02)
        for n in ("layer_norm1", "layer_norm2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
    sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
    if proj:
        sd["text_projection.weight"] = rn(proj, d, std=d ** -0.5)
    return sd


# ----------------------------------------------------------------------------------------------
# I2VGen-XL UNet (BASELINE config #5): parameter inventory and random-init weights in diffusers' key scheme
def i2vgen_param_shapes(cfg) -> dict:
    s = {}
    ch, T, ic, cd = cfg.block_out_channels, cfg.time_embed_dim, cfg.in_channels, cfg.cross_dim
    """parameter inventory of diffusers' I2VGenXLUNet (structure documented in tweediemix_amd/i2vgen.py; 1,420,469,224 parameters)"""

    def conv(n, i, o, k=3):
        s[n + ".weight"], s[n + ".bias"] = (o, i, k, k), (o,)

    def lin(n, i, o, bias=True):
        s[n + ".weight"] = (o, i)
        if bias:
            s[n + ".bias"] = (o,)

    def vec2(n, c):
        s[n + ".weight"], s[n + ".bias"] = (c,), (c,)

    def resnet(n, ci, co):
        vec2(n + ".norm1", ci); conv(n + ".conv1", ci, co); lin(n + ".time_emb_proj", T, co)
        vec2(n + ".norm2", co); conv(n + ".conv2", co, co)
        if ci != co:
            conv(n + ".conv_shortcut", ci, co, 1)

    def temp_conv(n, c):
        for k, idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
            vec2(f"{n}.conv{k}.0", c)
            s[f"{n}.conv{k}.{idx}.weight"], s[f"{n}.conv{k}.{idx}.bias"] = (c, c, 3, 1, 1), (c,)

    def block(n, dim, cross):
        vec2(n + ".norm1", dim); vec2(n + ".norm2", dim); vec2(n + ".norm3", dim)
        for a, kd in (("attn1", dim), ("attn2", cross)):
            lin(f"{n}.{a}.to_q", dim, dim, False); lin(f"{n}.{a}.to_k", kd, dim, False); lin(f"{n}.{a}.to_v", kd, dim, False)
            lin(f"{n}.{a}.to_out.0", dim, dim)
        lin(n + ".ff.net.0.proj", dim, 8 * dim); lin(n + ".ff.net.2", 4 * dim, dim)

    def t2d(n, c):
        vec2(n + ".norm", c); lin(n + ".proj_in", c, c); block(n + ".transformer_blocks.0", c, cd); lin(n + ".proj_out", c, c)

    def ttemp(n, c, inner):
        vec2(n + ".norm", c); lin(n + ".proj_in", c, inner); block(n + ".transformer_blocks.0", inner, inner); lin(n + ".proj_out", inner, c)

    conv("conv_in", 2 * ic, ch[0])
    ttemp("transformer_in", ch[0], cfg.transformer_in_heads * cfg.head_dim)
    conv("image_latents_proj_in.0", 4, ic * 4); conv("image_latents_proj_in.2", ic * 4, ic * 4); conv("image_latents_proj_in.4", ic * 4, ic)
    e = "image_latents_temporal_encoder"
    vec2(e + ".norm1", ic)
    lin(e + ".attn1.to_q", ic, 2 * ic, False); lin(e + ".attn1.to_k", ic, 2 * ic, False); lin(e + ".attn1.to_v", ic, 2 * ic, False)
    lin(e + ".attn1.to_out.0", 2 * ic, ic)
    lin(e + ".ff.net.0.proj", ic, ic * 4); lin(e + ".ff.net.2", ic * 4, ic)
    conv("image_latents_context_embedding.0", 4, ic * 8); conv("image_latents_context_embedding.3", ic * 8, ic * 16)
    conv("image_latents_context_embedding.5", ic * 16, cd)
    lin("time_embedding.linear_1", ch[0], T); lin("time_embedding.linear_2", T, T)
    lin("context_embedding.0", cd, T); lin("context_embedding.2", T, cd * ic)
    lin("fps_embedding.0", ch[0], T); lin("fps_embedding.2", T, T)
    nb = len(ch)
    ci = ch[0]
    skips = [ch[0]]
    for bi, co in enumerate(ch):
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{bi}.resnets.{j}", ci, co); temp_conv(f"down_blocks.{bi}.temp_convs.{j}", co)
            if cfg.attn_levels[bi]:
                t2d(f"down_blocks.{bi}.attentions.{j}", co); ttemp(f"down_blocks.{bi}.temp_attentions.{j}", co, co)
            ci = co
            skips.append(co)
        if bi < nb - 1:
            conv(f"down_blocks.{bi}.downsamplers.0.conv", co, co)
            skips.append(co)
    cm = ch[-1]
    resnet("mid_block.resnets.0", cm, cm); temp_conv("mid_block.temp_convs.0", cm)
    t2d("mid_block.attentions.0", cm); ttemp("mid_block.temp_attentions.0", cm, cm)
    resnet("mid_block.resnets.1", cm, cm); temp_conv("mid_block.temp_convs.1", cm)
    for ui in range(nb):
        bi = nb - 1 - ui
        co = ch[bi]
        for j in range(cfg.layers_per_block + 1):
            cs = skips.pop()
            resnet(f"up_blocks.{ui}.resnets.{j}", ci + cs, co); temp_conv(f"up_blocks.{ui}.temp_convs.{j}", co)
            if cfg.attn_levels[bi]:
                t2d(f"up_blocks.{ui}.attentions.{j}", co); ttemp(f"up_blocks.{ui}.temp_attentions.{j}", co, co)
            ci = co
        if ui < nb - 1:
            conv(f"up_blocks.{ui}.upsamplers.0.conv", co, co)
    vec2("conv_norm_out", ch[0]); conv("conv_out", ch[0], cfg.out_channels)
    return s


def synthetic_i2vgen_state_dict(cfg, seed=99, dtype=torch.float32, device="cpu"):
    """device="cuda": drawn on the device (1.42 B values: 24 s on the host cores, < 1 s there; a different stream of values than the CPU generator's)"""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shp in i2vgen_param_shapes(cfg).items():
        if name.endswith(".bias"):
            v = torch.randn(shp, generator=g, device=device) * 0.02
        elif len(shp) == 1:
            v = 1 + torch.randn(shp, generator=g, device=device) * 0.05
        else:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            v = torch.randn(shp, generator=g, device=device) * fan_in ** -0.5
            if name.endswith("conv4.3.weight"):
                v = v * 0.3                                   # (zero-init in diffusers; small but non-zero so tests exercise it)
        sd[name] = v.to(dtype)
    return sd




def synthetic_clip_vision_state_dict(d: int, layers: int, inter: int, heads: int, patch: int = 14, image: int = 224, proj: int = 1024,
                                     seed: int = 78, dtype=torch.float32):
    """random-init CLIP image tower in transformers' CLIPVisionModelWithProjection key scheme."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(dtype)
    n = (image // patch) ** 2 + 1
    sd = {"vision_model.embeddings.class_embedding": rn(d, std=0.02),
          "vision_model.embeddings.patch_embedding.weight": rn(d, 3, patch, patch, std=(3 * patch * patch) ** -0.5),
          "vision_model.embeddings.position_embedding.weight": rn(n, d, std=0.02)}
    for nm in ("pre_layrnorm", "post_layernorm"):
        sd[f"vision_model.{nm}.weight"], sd[f"vision_model.{nm}.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
    for i in range(layers):
        p = f"vision_model.encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"], sd[p + f"self_attn.{nm}.bias"] = rn(d, d, std=d ** -0.5), rn(d, std=0.02)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rn(inter, d, std=d ** -0.5), rn(inter, std=0.02)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rn(d, inter, std=inter ** -0.5), rn(d, std=0.02)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
    sd["visual_projection.weight"] = rn(proj, d, std=d ** -0.5)
    return sd
