"""Generates tests/golden/clip_tok/{vocab.json,merges.txt,cases.json}: a small synthetic CLIP byte-level BPE vocabulary
(512 byte tokens, merges learned on a toy corpus, the two special tokens) and the ids transformers' CLIPTokenizer (the
reference's tokenizer, fusion_sampling.py:27-41 `tokenize_prompt`: padding='max_length', truncation, 77 tokens) produces
for a set of prompts, with and without an added modifier token and with both pad conventions of the SDXL checkpoint
(tokenizer: pad = <|endoftext|>, tokenizer_2: pad = '!').  Run in the build container: python oracle/gen_golden_tokenizer.py"""
import collections, json, os
from transformers import CLIPTokenizer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "tests", "golden", "clip_tok")
os.makedirs(out, exist_ok=True)


def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


vocab = list(bytes_to_unicode().values())
vocab = vocab + [v + "</w>" for v in vocab]
corpus = ("a photo of a cat and a dog playing on the beach , a cat wearing sunglasses . the dog is running in a field of "
          "flowers ! photo photos cats dogs mountain mountains in the background a teddy bear sitting on a wooden chair "
          "it's the dog's photo with a pink flower and a blue sky").split()
ws = {tuple(list(w[:-1]) + [w[-1] + "</w>"]): c for w, c in collections.Counter(corpus).items()}
merges = []
for _ in range(120):
    pc = collections.Counter()
    for w, c in ws.items():
        for p in zip(w[:-1], w[1:]):
            pc[p] += c
    if not pc:
        break
    best = max(sorted(pc), key=lambda p: pc[p])
    merges.append(best)
    nws = {}
    for w, c in ws.items():
        o, i = [], 0
        while i < len(w):
            if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                o.append(w[i] + w[i + 1]); i += 2
            else:
                o.append(w[i]); i += 1
        nws[tuple(o)] = c
    ws = nws
vocab += [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
json.dump({t: i for i, t in enumerate(vocab)}, open(os.path.join(out, "vocab.json"), "w"))
open(os.path.join(out, "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")

texts = ["A photo of a cat and a dog", "a  <new1> cat wearing sunglasses, in the mountains!", "it's the dog's photo 42",
         "", "a " * 100, "a <new2> teddy bear sitting on a wooden chair, <new1> dog in the background",
         "Café au lait — naïve résumé #1 (photo)", "  leading and   trailing   spaces  ", "blurry, low quality"]
cases = []
for pad in ("<|endoftext|>", "!"):
    tok = CLIPTokenizer(os.path.join(out, "vocab.json"), os.path.join(out, "merges.txt"), pad_token=pad)
    ids0 = tok(texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids.tolist()
    added = [tok.add_tokens(t) for t in ("<new1>", "<new2>")]
    new_ids = [tok.convert_tokens_to_ids(t) for t in ("<new1>", "<new2>")]
    ids1 = tok(texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids.tolist()
    cases.append({"pad_token": pad, "texts": texts, "ids_plain": ids0, "added": ["<new1>", "<new2>"], "added_ids": new_ids,
                  "len_after": len(tok), "ids_added": ids1, "tokens_added": [tok.tokenize(t) for t in texts]})
json.dump(cases, open(os.path.join(out, "cases.json"), "w"))
print("vocab", len(vocab), "merges", len(merges), "cases", len(cases))
