"""CPU: the C-ABI library loads here (no GPU) and exports every symbol include/tmix.h declares;
the ctypes mirrors of the descriptor structs have the C layout."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "tmix.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tmix_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from tweediemix_amd import lib
    l = lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/tmix.h but not exported"
        assert s in lib.SIGNATURES, f"{s} has no ctypes prototype"
    assert l.tmix_version() == 100
    assert isinstance(l.tmix_last_error_string(), bytes)


def test_descriptor_layouts():
    from tweediemix_amd import lib
    # tmix_gemm_desc: 8-byte aligned fields in declaration order (see include/tmix.h)
    assert lib.GemmDesc.A.offset == 0 and lib.GemmDesc.W.offset == 24 and lib.GemmDesc.C.offset == 48
    assert lib.GemmDesc.bias.offset == 72 and lib.GemmDesc.residual.offset == 88
    assert lib.GemmDesc.rowgroup_bias.offset == 112 and lib.GemmDesc.rows_per_group.offset == 120
    assert lib.GemmDesc.Ct.offset == 128 and lib.GemmDesc.n_trans_begin.offset == 152
    assert lib.GemmDesc.M.offset == 156 and lib.GemmDesc.epilogue.offset == 172 and lib.GemmDesc.tile_cfg.offset == 176
    assert lib.GemmDesc.row_stats_out.offset == 184 and lib.GemmDesc.ln_stats.offset == 208 and lib.GemmDesc.ln_colsum.offset == 232
    assert lib.GemmDesc.ln_inv_c.offset == 248 and lib.GemmDesc.ln_parts.offset == 256 and ctypes.sizeof(lib.GemmDesc) == 264
    assert lib.ConvDesc.B.offset == 48 and lib.ConvDesc.tile_cfg.offset == 72 and ctypes.sizeof(lib.ConvDesc) == 80


def test_geometry_helper_without_gpu():
    from tweediemix_amd import lib
    l = lib.load()
    assert l.tmix_groupnorm_ws_chunks(16384) == 128 and l.tmix_groupnorm_ws_chunks(16) == 1


def test_ops_refuse_cpu_tensors():
    import torch
    from tweediemix_amd import lib, ops
    with pytest.raises(lib.TmixError):
        ops.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))
