"""Generates tests/golden/prompts.json by EXECUTING the reference's own prompt-assembly and modifier-token-injection
statements (fusion_generation/fusion_sampling.py:139-154 and :159-190, read from /root/reference at generation time only)
on several argument sets, with tiny stand-ins for the tokenizer / embedding-table objects those statements touch.
The fixture holds inputs and outputs only.  Run in the build container: python oracle/gen_golden_prompts.py"""
import json, os, textwrap, types
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open("/root/reference/fusion_generation/fusion_sampling.py").read().split("\n")
assemble = textwrap.dedent("\n".join(lines[138:154]))
inject = textwrap.dedent("\n".join(lines[158:190]))


class Tok:
    def __init__(self, n):
        self.n, self.added = n, {}
    def add_tokens(self, t):
        if t in self.added:
            return 0
        self.added[t] = self.n + len(self.added)
        return 1
    def convert_tokens_to_ids(self, t):
        return self.added[t]
    def __len__(self):
        return self.n + len(self.added)


class Enc:
    def __init__(self, n, d):
        self.w = types.SimpleNamespace(weight=types.SimpleNamespace(data=torch.zeros(n, d)))
    def resize_token_embeddings(self, n):
        old = self.w.weight.data
        self.w.weight.data = torch.cat([old, torch.zeros(n - old.shape[0], old.shape[1])])
    def get_input_embeddings(self):
        return self.w


cases = [
    dict(prompt="a photo of a cat wearing sunglasses+a photo of a dog in a field+a photo of a beach",
         prompt_orig="a cat and a dog on the beach", concepts="cat+dog+beach", modifier_token="<new1>+<new2>+<new3>"),
    dict(prompt="a teddy bear sitting+a wooden chair+garden in the background", prompt_orig="a teddy on a chair+ignored",
         concepts="teddy+chair+garden", modifier_token="<t1>+<t2>+<t3>"),
    dict(prompt="a dog+a cat", prompt_orig="dog and cat", concepts="dog+zebra", modifier_token="<a>+<b>"),   # word not found: find() == -1
]
out = []
for c in cases:
    config = types.SimpleNamespace(personal_checkpoint="x.bin+y.bin+z.bin", **c)
    self = types.SimpleNamespace(sts=[])
    env = {"config": config, "self": self}
    exec(assemble, env)
    K = env["concept_num"]
    g = torch.Generator().manual_seed(3)
    self.sts = [{"modifier_token": {f"<ck{i}>": torch.randn(8, generator=g)}, "modifier_token_2": {f"<ck{i}>": torch.randn(12, generator=g)},
                 "unet": {}} for i in range(K)]
    self.tokenizer, self.tokenizer_2 = Tok(50), Tok(60)
    self.text_encoder, self.text_encoder_2 = Enc(50, 8), Enc(60, 12)
    env2 = {"self": self, "modifier_token_user": env["modifier_token_user"], "torch": torch}
    exec(inject, env2)
    out.append(dict(args=c, prompts=env["prompts"], prompts_single=env["prompts_single"], concept_num=K,
                    ckpt_tokens=[[list(st["modifier_token"].keys())[0], st["modifier_token"][f"<ck{i}>"].tolist(),
                                  st["modifier_token_2"][f"<ck{i}>"].tolist()] for i, st in enumerate(self.sts)],
                    ids=env2["modifier_token_id"], ids_2=env2["modifier_token_id_2"],
                    table_rows=self.text_encoder.get_input_embeddings().weight.data[50:].tolist(),
                    table_rows_2=self.text_encoder_2.get_input_embeddings().weight.data[60:].tolist()))
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "prompts.json"), "w"), indent=0)
print(json.dumps([o["prompts"] for o in out], indent=1))
