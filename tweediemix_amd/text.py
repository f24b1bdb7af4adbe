"""Init-time text path of the sampler: what `Tweediemix.__init__` does with the two SDXL text
encoders before the denoising loop starts (fusion_generation/fusion_sampling.py:139-196 prompt assembly + modifier-token
injection, :43-68 `encode_prompt`, :268-285 `get_text_embeds`), on this package's HIP kernels:

  CLIP ViT-L/14 text tower  (transformers CLIPTextModel:               12 layers, d=768,  12 heads, quick_gelu)
  OpenCLIP bigG/14 tower    (transformers CLIPTextModelWithProjection: 32 layers, d=1280, 20 heads, gelu, text_projection)

per layer: tmix_layernorm -> QKV GEMM (V stored transposed) -> per prompt: QK^T GEMM with fp32 scores (heads as the batch
dimension) -> tmix_softmax_rows_causal -> PV GEMM -> out-proj GEMM (+residual) -> tmix_layernorm -> fc1 GEMM with the
activation in its epilogue -> fc2 GEMM (+residual).  Output = hidden_states[-2] of both towers concatenated ([P,77,2048])
and the projected pooled row of the second tower ([P,1280]).  The tokenizer is a from-scratch CLIP byte-level BPE reading
the checkpoint's vocab.json / merges.txt."""
from __future__ import annotations

import json
import os

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


# ============================================================================================ encoder
class ClipTextEncoder:
    """one CLIP text tower on the HIP kernels.  sd: transformers state dict ('text_model.' prefix optional)."""

    KPAD = 128                      # key axis padded to a multiple of the GEMM K-tile (64) for the PV product

    def __init__(self, sd: dict, heads: int, act: str = "quick_gelu", eos_token_id: int = 2, eps: float = 1e-5, device="cuda"):
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
        self.dev = torch.device(device)
        self.heads, self.act, self.eos, self.eps = heads, act, eos_token_id, eps
        assert act in ("quick_gelu", "gelu")
        f32 = lambda k: sd[k].to(self.dev, F32).contiguous()
        bf = lambda t: t.to(self.dev, BF16).contiguous()
        self.tok = f32("embeddings.token_embedding.weight")           # fp32 table: rows get overwritten by modifier tokens
        self.pos = f32("embeddings.position_embedding.weight")
        self.d = self.tok.shape[1]
        assert self.d % heads == 0 and self.d // heads == 64, "head_dim 64 (both SDXL towers)"
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        self.layers = []
        for i in range(self.n_layers):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            self.layers.append(dict(
                ln1=(f32(p + "layer_norm1.weight"), f32(p + "layer_norm1.bias")),
                qkv=bf(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]])),
                qkv_b=torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]]).to(self.dev, F32).contiguous(),
                out=bf(sd[a + "out_proj.weight"]), out_b=f32(a + "out_proj.bias"),
                ln2=(f32(p + "layer_norm2.weight"), f32(p + "layer_norm2.bias")),
                fc1=bf(sd[p + "mlp.fc1.weight"]), fc1_b=f32(p + "mlp.fc1.bias"),
                fc2=bf(sd[p + "mlp.fc2.weight"]), fc2_b=f32(p + "mlp.fc2.bias")))
        self.final_ln = (f32("final_layer_norm.weight"), f32("final_layer_norm.bias"))
        self.proj = bf(sd["text_projection.weight"]) if "text_projection.weight" in sd else None

    # -- vocabulary surgery of fusion_sampling.py:161-189 (resize_token_embeddings + row overwrite)
    def resize_token_embeddings(self, n: int):
        if n > self.tok.shape[0]:
            extra = torch.zeros(n - self.tok.shape[0], self.d, device=self.dev, dtype=F32)
            self.tok = torch.cat([self.tok, extra]).contiguous()

    def set_token_embedding(self, token_id: int, vec: torch.Tensor):
        self.tok[token_id] = vec.to(self.dev, F32)

    # -- forward
    def _layer(self, x, lw, B, S):
        d, H, KP = self.d, self.heads, self.KPAD
        M = B * S
        y = ops.layernorm(x, lw["ln1"][0], lw["ln1"][1], self.eps)
        qk = torch.zeros(M + 8, 2 * d, device=self.dev, dtype=BF16)                 # +8 rows: the score GEMM reads K rows up to 80
        vt = torch.zeros(B, d, KP, device=self.dev, dtype=BF16)
        ops.gemm(y.view(B, S, d), lw["qkv"], out=qk[:M].view(B, S, 2 * d), bias=lw["qkv_b"], out_t=vt, n_trans_begin=2 * d)
        ao = torch.empty(M, d, device=self.dev, dtype=BF16)
        scores = torch.empty(H, S, KP, device=self.dev, dtype=F32)
        probs = torch.empty(H, S, KP, device=self.dev, dtype=BF16)
        NK = (S + 3) // 4 * 4
        for b in range(B):
            q = qk[b * S:b * S + S].as_strided((H, S, 64), (64, 2 * d, 1))
            k = qk[b * S:].as_strided((H, NK, 64), (64, 2 * d, 1), storage_offset=qk[b * S:].storage_offset() + d)
            ops.gemm(q, k, out_f32=scores[:, :, :NK])
            ops.softmax_rows_causal(scores.view(H * S, KP), probs.view(H * S, KP), S, 64 ** -0.5)
            o = ao[b * S:b * S + S].as_strided((H, S, 64), (64, d, 1))
            ops.gemm(probs, vt[b].view(H, 64, KP), out=o)
        x = ops.gemm(ao, lw["out"], bias=lw["out_b"], residual=x)
        y = ops.layernorm(x, lw["ln2"][0], lw["ln2"][1], self.eps)
        h = ops.gemm(y, lw["fc1"], bias=lw["fc1_b"], act=self.act)
        return ops.gemm(h, lw["fc2"], bias=lw["fc2_b"], residual=x)

    @torch.no_grad()
    def last_hidden_state(self, input_ids: torch.Tensor):
        """[B,S,d] bf16 output of the full tower WITH the final LayerNorm: `text_encoder(ids)[0]`, what the I2VGen-XL pipeline
        feeds its UNet (video_gen/pipeline_i2vgen_xl.py encode_prompt) -- the SDXL samplers use hidden_states[-2] instead."""
        ids = input_ids.to(self.dev).long()
        B, S = ids.shape
        x = (self.tok[ids] + self.pos[:S][None]).to(BF16).reshape(B * S, self.d).contiguous()
        for lw in self.layers:
            x = self._layer(x, lw, B, S)
        return ops.layernorm(x, self.final_ln[0], self.final_ln[1], self.eps).view(B, S, self.d)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, need_pooled: bool = True):
        """input_ids [B,S] -> (hidden_states[-2] [B,S,d] bf16, pooled [B,proj_dim or d] fp32 or None)."""
        ids = input_ids.to(self.dev).long()
        B, S = ids.shape
        assert S <= self.KPAD and S <= self.pos.shape[0]
        x = (self.tok[ids] + self.pos[:S][None]).to(BF16).reshape(B * S, self.d).contiguous()
        for lw in self.layers[:-1]:
            x = self._layer(x, lw, B, S)
        hs_m2 = x.view(B, S, self.d)
        if not need_pooled:
            return hs_m2, None
        x = self._layer(x, self.layers[-1], B, S)
        last = ops.layernorm(x, self.final_ln[0], self.final_ln[1], self.eps).view(B, S, self.d)
        # transformers modeling_clip.py: legacy configs (eos_token_id == 2) pool at the LARGEST id, others at the first EOS
        pos = ids.argmax(-1) if self.eos == 2 else (ids == self.eos).int().argmax(-1)
        pooled = last[torch.arange(B, device=self.dev), pos].float().contiguous()
        if self.proj is not None:
            pooled = ops.linear_small(pooled, self.proj)
        return hs_m2, pooled


def encode_prompt(encoders, ids_list):
    """fusion_sampling.py:43-68: hidden_states[-2] of every tower concatenated on the feature axis; the pooled output
    is that of the LAST tower.  Returns (prompt_embeds [P,77,sum d] bf16, pooled [P,1280] fp32)."""
    embeds, pooled = [], None
    for n, (enc, ids) in enumerate(zip(encoders, ids_list)):
        hs, p = enc(ids, need_pooled=(n == len(encoders) - 1))
        embeds.append(hs)
        pooled = p if p is not None else pooled
    return torch.cat(embeds, dim=-1), pooled


# ============================================================================================ tokenizer
def _bytes_to_unicode():
    """the reversible byte -> printable-unicode map of GPT-2 / CLIP byte-level BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {b: chr(c) for b, c in zip(bs, cs)}


class ClipBPETokenizer:
    """CLIP's lower-casing byte-level BPE (the `CLIPTokenizer` the reference takes from the SDXL checkpoint's
    tokenizer/ and tokenizer_2/ folders; `tokenize_prompt`, fusion_sampling.py:27-41: pad to 77, truncate, ids only).
    Supports `add_tokens` / `convert_tokens_to_ids` / `len()` as used for the modifier tokens (:161-175)."""

    PATTERN = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"

    def __init__(self, vocab: dict, merges: list, pad_token: str = "<|endoftext|>", model_max_length: int = 77,
                 bos_token: str = "<|startoftext|>", eos_token: str = "<|endoftext|>"):
        import regex
        self.encoder = dict(vocab)
        self.ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.byte_map = _bytes_to_unicode()
        self.pat = regex.compile(self.PATTERN, regex.IGNORECASE)
        self.bos_token, self.eos_token, self.pad_token = bos_token, eos_token, pad_token
        self.model_max_length = model_max_length
        self.added = {}
        self.cache = {}

    @classmethod
    def from_pretrained(cls, path: str):
        """path: a tokenizer folder of the SDXL checkpoint (vocab.json, merges.txt, special_tokens_map.json)."""
        with open(os.path.join(path, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        with open(os.path.join(path, "merges.txt"), encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(l.split()) for l in lines[1:] if l.strip() and len(l.split()) == 2]
        pad, maxlen = "<|endoftext|>", 77
        for name in ("special_tokens_map.json", "tokenizer_config.json"):
            fp = os.path.join(path, name)
            if os.path.exists(fp):
                with open(fp, encoding="utf-8") as f:
                    cfg = json.load(f)
                p = cfg.get("pad_token")
                if p is not None:
                    pad = p["content"] if isinstance(p, dict) else p
                if isinstance(cfg.get("model_max_length"), int) and cfg["model_max_length"] < 100000:
                    maxlen = cfg["model_max_length"]
        return cls(vocab, merges, pad_token=pad, model_max_length=maxlen)

    def __len__(self):
        return len(self.encoder) + len(self.added)

    def add_tokens(self, token: str) -> int:
        if token in self.encoder or token in self.added:
            return 0
        self.added[token] = len(self)
        return 1

    def convert_tokens_to_ids(self, token: str) -> int:
        return self.added[token] if token in self.added else self.encoder[token]

    @property
    def bos_token_id(self):
        return self.encoder[self.bos_token]

    @property
    def eos_token_id(self):
        return self.encoder[self.eos_token]

    @property
    def pad_token_id(self):
        return self.convert_tokens_to_ids(self.pad_token)

    def _bpe(self, token: str):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = set(zip(word[:-1], word[1:]))
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
        self.cache[token] = word
        return word

    def _tokenize_plain(self, text: str):
        import unicodedata
        import regex
        text = regex.sub(r"\s+", " ", unicodedata.normalize("NFC", text)).strip().lower()
        out = []
        for tok in self.pat.findall(text):
            if tok in (self.bos_token, self.eos_token):
                out.append(tok)
                continue
            mapped = "".join(self.byte_map[b] for b in tok.encode("utf-8"))
            out.extend(self._bpe(mapped))
        return out

    def tokenize(self, text: str):
        """added and special tokens are cut out of the raw text first (longest first); the pieces in between go through
        BPE.  (tokenizer_2 of the SDXL checkpoint pads with '!', so a literal '!' in a prompt becomes the pad id 0, not
        '!</w>' -- same as the reference's tokenizer.)"""
        import regex
        cut = set(self.added) | {self.bos_token, self.eos_token, self.pad_token}
        alts = "|".join(regex.escape(t) for t in sorted(cut, key=len, reverse=True))
        out = []
        for piece in regex.split(f"({alts})", text):
            if piece in cut:
                out.append(piece)
            elif piece:
                out.extend(self._tokenize_plain(piece))
        return out

    def __call__(self, prompts, max_length: int | None = None) -> torch.Tensor:
        """ids [P, max_length]: <bos> tokens (truncated) <eos> then pad -- padding='max_length', truncation=True."""
        if isinstance(prompts, str):
            prompts = [prompts]
        n = max_length or self.model_max_length
        rows = []
        for p in prompts:
            ids = [self.convert_tokens_to_ids(t) for t in self.tokenize(p)][:n - 2]
            ids = [self.bos_token_id] + ids + [self.eos_token_id]
            rows.append(ids + [self.pad_token_id] * (n - len(ids)))
        return torch.tensor(rows, dtype=torch.long)


# ============================================================================================ prompt plumbing
def assemble_prompts(prompt: str, prompt_orig: str, concepts: str, modifier_token: str):
    """fusion_sampling.py:139-156.  Returns (prompts, prompts_single, concept_num): prompts[0] is the full-scene prompt,
    prompts[1+i] is the i-th single-concept prompt with its modifier token put in front of the concept word
    (`str.find` semantics kept: a concept word that does not occur gives index -1, i.e. the token lands before the last
    character, exactly like the reference); prompts_single are the first concept_num-1 un-modified prompts."""
    prompt_sep = prompt.split('+')
    concept_list = concepts.split('+')
    modifier_token_user = modifier_token.split('+')
    prompts = [prompt_orig.split('+')[0]]
    concept_num = len(concept_list)
    prompts_single = prompt_sep[:concept_num - 1]
    for i, wd in enumerate(concept_list):
        index = prompt_sep[i].find(wd)
        prompts.append(prompt_sep[i][:index] + modifier_token_user[i] + " " + prompt_sep[i][index:])
    return prompts, prompts_single, concept_num


def inject_modifier_tokens(tokenizers, encoders, sts, modifier_token_user):
    """fusion_sampling.py:158-189: every user modifier token is appended to both vocabularies, both embedding tables
    grow, and row id_i receives checkpoint i's learned embedding (keys 'modifier_token' / 'modifier_token_2').
    Like the reference, the embedding for the i-th user token is `sts[i][...][keys[i]]` with keys = the concatenation of
    every checkpoint's token names, and nothing happens unless the FIRST checkpoint carries 'modifier_token'."""
    if not sts or 'modifier_token' not in sts[0]:
        return [], []
    keys, keys_2 = [], []
    for st in sts:
        keys += list(st['modifier_token'].keys())
        keys_2 += list(st['modifier_token_2'].keys())
    ids, ids_2 = [], []
    for tok in modifier_token_user:
        tokenizers[0].add_tokens(tok)
        ids.append(tokenizers[0].convert_tokens_to_ids(tok))
        tokenizers[1].add_tokens(tok)
        ids_2.append(tokenizers[1].convert_tokens_to_ids(tok))
    encoders[0].resize_token_embeddings(len(tokenizers[0]))
    encoders[1].resize_token_embeddings(len(tokenizers[1]))
    for i, id_ in enumerate(ids):
        encoders[0].set_token_embedding(id_, sts[i]['modifier_token'][keys[i]])
    for i, id_ in enumerate(ids_2):
        encoders[1].set_token_embedding(id_, sts[i]['modifier_token_2'][keys_2[i]])
    return ids, ids_2


def get_text_embeds(encoders, tokenizers, prompt, negative_prompt):
    """fusion_sampling.py:268-285: rows = [negative prompt(s), prompts...] -> (embeds [1+P,77,2048], pooled [1+P,1280])."""
    pe, pp = encode_prompt(encoders, [t(prompt) for t in tokenizers])
    ue, up = encode_prompt(encoders, [t(negative_prompt) for t in tokenizers])
    return torch.cat([ue, pe]), torch.cat([up, pp])


def load_text_tower(folder: str, device="cuda") -> ClipTextEncoder:
    """a text_encoder/ or text_encoder_2/ folder of a diffusers-layout SDXL checkpoint (config.json + safetensors)."""
    with open(os.path.join(folder, "config.json")) as f:
        cfg = json.load(f)
    sd = None
    for name in ("model.fp16.safetensors", "model.safetensors"):
        fp = os.path.join(folder, name)
        if os.path.exists(fp):
            from safetensors.torch import load_file
            sd = load_file(fp)
            break
    if sd is None:
        sd = torch.load(os.path.join(folder, "pytorch_model.bin"), map_location="cpu")
    act = cfg.get("hidden_act", "quick_gelu")
    if act not in ("quick_gelu", "gelu"):
        raise ValueError(f"text tower activation {act!r} is not one of the two SDXL uses")
    return ClipTextEncoder(sd, cfg["num_attention_heads"], act, cfg.get("eos_token_id", 2), cfg.get("layer_norm_eps", 1e-5), device)


class TextPath:
    """the text half of `Tweediemix.__init__` (fusion_sampling.py:139-196) for a local SDXL checkpoint folder."""

    def __init__(self, sd_path: str, device="cuda"):
        self.tokenizers = [ClipBPETokenizer.from_pretrained(os.path.join(sd_path, "tokenizer")),
                           ClipBPETokenizer.from_pretrained(os.path.join(sd_path, "tokenizer_2"))]
        self.encoders = [load_text_tower(os.path.join(sd_path, "text_encoder"), device),
                         load_text_tower(os.path.join(sd_path, "text_encoder_2"), device)]

    def embed(self, opt, sts):
        """-> (text_embeds, text_embeds_single, concept_num) as the sampler takes them."""
        prompts, prompts_single, K = assemble_prompts(opt.prompt, opt.prompt_orig, opt.concepts, opt.modifier_token)
        inject_modifier_tokens(self.tokenizers, self.encoders, sts, opt.modifier_token.split('+'))
        null = [opt.negative_prompt]
        return (get_text_embeds(self.encoders, self.tokenizers, prompts, null),
                get_text_embeds(self.encoders, self.tokenizers, prompts_single, null), K)


# ============================================================================================ vision tower
class ClipVisionEncoder:
    """CLIP image tower (transformers CLIPVisionModelWithProjection; OpenCLIP ViT-H/14 in the I2VGen-XL pipeline's `_encode_image`:
    32 layers, d=1280, 16 heads of 80) on the HIP kernels.  The patch convolution is a GEMM over unfolded patches (K padded to a
    multiple of 64); heads of 80 are zero-padded to 128 columns in the fused q/k projection so the score GEMM has a legal K;
    scores / probabilities live in rows padded to a multiple of 64 keys (`tmix_softmax_rows_masked` zeroes the padding)."""

    HP = 128                         # padded head size of q / k

    def __init__(self, sd: dict, heads: int, patch: int, act: str = "gelu", eps: float = 1e-5, device="cuda"):
        sd = {(k[len("vision_model."):] if k.startswith("vision_model.") else k): v for k, v in sd.items()}
        self.dev, self.heads, self.patch, self.act, self.eps = torch.device(device), heads, patch, act, eps
        dev = self.dev
        f32 = lambda k: sd[k].to(dev, F32).contiguous()
        bf = lambda t: t.to(dev, BF16).contiguous()
        wp = sd["embeddings.patch_embedding.weight"].to(dev, F32)                    # [d, 3, p, p]
        self.d = d = wp.shape[0]
        self.hd = d // heads
        assert self.hd <= self.HP and self.hd % 4 == 0
        k0 = wp[0].numel()
        self.kp = (k0 + 63) // 64 * 64
        wpad = torch.zeros(d, self.kp, device=dev, dtype=F32)
        wpad[:, :k0] = wp.reshape(d, k0)
        self.patch_w = bf(wpad)
        self.cls, self.pos = f32("embeddings.class_embedding"), f32("embeddings.position_embedding.weight")
        self.pre_ln = (f32("pre_layrnorm.weight"), f32("pre_layrnorm.bias"))
        self.post_ln = (f32("post_layernorm.weight"), f32("post_layernorm.bias"))
        self.proj = bf(sd["visual_projection.weight"])
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
        HP, hd = self.HP, self.hd

        def pad_heads(w, b):         # [d, d] / [d] -> [heads*HP, d] / [heads*HP], rows h*HP .. h*HP+hd-1 carry head h
            wo = torch.zeros(heads * HP, d, device=dev, dtype=F32)
            bo = torch.zeros(heads * HP, device=dev, dtype=F32)
            for h in range(heads):
                wo[h * HP:h * HP + hd] = w[h * hd:(h + 1) * hd]
                bo[h * HP:h * HP + hd] = b[h * hd:(h + 1) * hd]
            return wo, bo

        self.layers = []
        for i in range(self.n_layers):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            wq, bq = pad_heads(sd[a + "q_proj.weight"].to(dev, F32), sd[a + "q_proj.bias"].to(dev, F32))
            wk, bk = pad_heads(sd[a + "k_proj.weight"].to(dev, F32), sd[a + "k_proj.bias"].to(dev, F32))
            self.layers.append(dict(
                ln1=(f32(p + "layer_norm1.weight"), f32(p + "layer_norm1.bias")),
                qkv=bf(torch.cat([wq, wk, sd[a + "v_proj.weight"].to(dev, F32)])),
                qkv_b=torch.cat([bq, bk, sd[a + "v_proj.bias"].to(dev, F32)]).contiguous(),
                out=bf(sd[a + "out_proj.weight"]), out_b=f32(a + "out_proj.bias"),
                ln2=(f32(p + "layer_norm2.weight"), f32(p + "layer_norm2.bias")),
                fc1=bf(sd[p + "mlp.fc1.weight"]), fc1_b=f32(p + "mlp.fc1.bias"),
                fc2=bf(sd[p + "mlp.fc2.weight"]), fc2_b=f32(p + "mlp.fc2.bias")))

    def _layer(self, x, lw, B, S):
        d, H, HP, hd = self.d, self.heads, self.HP, self.hd
        M = B * S
        KP = (S + 63) // 64 * 64                                      # padded key axis (K of the PV product)
        NK = (S + 3) // 4 * 4
        QK = 2 * H * HP
        y = ops.layernorm(x, lw["ln1"][0], lw["ln1"][1], self.eps)
        qk = torch.zeros(M + 8, QK, device=self.dev, dtype=BF16)
        vt = torch.zeros(B, d, KP, device=self.dev, dtype=BF16)
        ops.gemm(y.view(B, S, d), lw["qkv"], out=qk[:M].view(B, S, QK), bias=lw["qkv_b"], out_t=vt, n_trans_begin=QK)
        ao = torch.empty(M, d, device=self.dev, dtype=BF16)
        scores = torch.empty(H, S, KP, device=self.dev, dtype=F32)
        probs = torch.empty(H, S, KP, device=self.dev, dtype=BF16)
        for b in range(B):
            q = qk[b * S:b * S + S].as_strided((H, S, HP), (HP, QK, 1))
            k = qk[b * S:].as_strided((H, NK, HP), (HP, QK, 1), storage_offset=qk[b * S:].storage_offset() + H * HP)
            ops.gemm(q, k, out_f32=scores[:, :, :NK])
            ops.softmax_rows_masked(scores.view(H * S, KP), probs.view(H * S, KP), S, hd ** -0.5)
            o = ao[b * S:b * S + S].as_strided((H, S, hd), (hd, d, 1))
            ops.gemm(probs, vt[b].view(H, hd, KP), out=o)
        x = ops.gemm(ao, lw["out"], bias=lw["out_b"], residual=x)
        y = ops.layernorm(x, lw["ln2"][0], lw["ln2"][1], self.eps)
        h = ops.gemm(y, lw["fc1"], bias=lw["fc1_b"], act=self.act)
        return ops.gemm(h, lw["fc2"], bias=lw["fc2_b"], residual=x)

    @torch.no_grad()
    def __call__(self, pixel_values: torch.Tensor):
        """pixel_values [B,3,H,W] (already normalised) -> image_embeds [B, projection_dim] fp32."""
        pv = pixel_values.to(self.dev, F32)
        B, _c, Hh, Ww = pv.shape
        p = self.patch
        gh, gw = Hh // p, Ww // p
        S = gh * gw + 1
        assert S == self.pos.shape[0], "position table is for a different image size"
        patches = pv.unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B * gh * gw, -1)      # [B*N, 3*p*p]
        a = torch.zeros(B * gh * gw, self.kp, device=self.dev, dtype=BF16)
        a[:, :patches.shape[1]] = patches.to(BF16)
        emb = ops.gemm(a, self.patch_w).view(B, gh * gw, self.d).float()
        x = torch.cat([self.cls.expand(B, 1, self.d), emb], dim=1) + self.pos[None]
        x = ops.layernorm(x.to(BF16).reshape(B * S, self.d).contiguous(), self.pre_ln[0], self.pre_ln[1], self.eps)
        for lw in self.layers:
            x = self._layer(x, lw, B, S)
        cls = x.view(B, S, self.d)[:, 0].contiguous()
        pooled = ops.layernorm(cls, self.post_ln[0], self.post_ln[1], self.eps).float().contiguous()
        return ops.linear_small(pooled, self.proj)
