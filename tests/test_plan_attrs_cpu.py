"""CPU: the video plan (I2VPlan) reuses the image UNet's op emitters without running UNetPlan.__init__, so every attribute those
emitters read through `self.` must be set by I2VPlan itself.  (Round 3 broke exactly that once: a new `self.lowrank` read in
UNetPlan._t2d made the video plan -- and with it the default bench line -- fail with AttributeError.)  Checked statically on the sources."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cls(path, name):
    tree = ast.parse(open(os.path.join(ROOT, path)).read())
    return next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == name)


def _self_reads(fn):
    """attribute names read as self.X inside fn (not via getattr(self, 'X', default), not assignment targets)"""
    stores = {id(n) for n in ast.walk(fn) if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store)}
    return {n.attr for n in ast.walk(fn) if isinstance(n, ast.Attribute) and id(n) not in stores
            and isinstance(n.value, ast.Name) and n.value.id == "self"}


def _self_writes(cls):
    out = set()
    for n in ast.walk(cls):
        if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store) and isinstance(n.value, ast.Name) and n.value.id == "self":
            out.add(n.attr)
    return out


def test_video_plan_sets_every_attribute_the_shared_emitters_read():
    unet = _cls("tweediemix_amd/unet.py", "UNetPlan")
    i2v = _cls("tweediemix_amd/i2vgen.py", "I2VPlan")
    methods = {n.name: n for n in unet.body if isinstance(n, ast.FunctionDef)}
    own = {n.name for n in i2v.body if isinstance(n, ast.FunctionDef)}
    shared = [m for name, m in methods.items() if name not in own and name not in ("__init__", "_build")]
    assert {"_gn", "_conv", "_gemm", "_t2d", "_proj", "_resnet"} <= {m.name for m in shared}, "the emitters the video plan borrows"
    # what the borrowed emitters can call on self: UNetPlan's methods / class attributes, I2VPlan's own, and everything I2VPlan assigns
    have = _self_writes(i2v) | set(methods) | own | {t.id for n in unet.body + i2v.body if isinstance(n, ast.Assign) for t in n.targets if isinstance(t, ast.Name)}
    # attributes a borrowed emitter itself creates before reading them (caches)
    for m in shared:
        have |= {n.attr for n in ast.walk(m) if isinstance(n, ast.Attribute) and isinstance(n.ctx, ast.Store) and isinstance(n.value, ast.Name) and n.value.id == "self"}
    missing = {}
    for m in shared:
        need = _self_reads(m) - have
        if need:
            missing[m.name] = sorted(need)
    assert not missing, f"UNetPlan emitters read attributes I2VPlan never sets: {missing}"
