"""CPU (no GPU needed): the C-ABI library builds, loads, and exports every symbol include/sam_hip.h declares; the
ctypes signatures cover the header; and the product path refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "sam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sam_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import sam_textvqa_amd._build as b
    lib_path = b.build()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsam_hip.so does not export %s declared in include/sam_hip.h" % n


def test_ctypes_binding_covers_the_header():
    from sam_textvqa_amd import _capi
    declared = set(header_functions()) - {"sam_last_error", "sam_device_info", "sam_gemm_desc", "sam_build_digest"}
    bound = set(_capi.SIGNATURES)
    assert declared <= bound, "unbound entry points: %s" % sorted(declared - bound)
    assert bound <= set(header_functions()), "bound but undeclared: %s" % sorted(bound - set(header_functions()))
    l = _capi.lib()
    assert l.sam_abi_version() == 9
    import sam_textvqa_amd._build as b
    assert l.sam_build_digest().decode() == b._digest()          # the binary that is loaded is the one built from the sources in the tree
    assert _capi.call("sam_attn_words_per_row", 182) == 6 and _capi.call("sam_attn_words_per_row", 20) == 1
    assert _capi.call("sam_attn_words_per_row", 350) == 12 and _capi.call("sam_attn_words_per_row", 385) == -1
    # CUs withheld from persistent grids: a multiple of 8, bounded, host-side state only (no GPU needed)
    was = _capi.call("sam_get_cu_reserve")
    _capi.call("sam_set_cu_reserve", 37)
    assert _capi.call("sam_get_cu_reserve") == 32
    with pytest.raises(_capi.SamHipError):
        _capi.call("sam_set_cu_reserve", 200)
    _capi.call("sam_set_cu_reserve", was)


def test_gemm_desc_layout_matches_header_field_order():
    from sam_textvqa_amd import _capi
    src = open(os.path.join(ROOT, "include", "sam_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", src[src.index("typedef struct sam_gemm_desc {"): src.index("} sam_gemm_desc;")], flags=re.S)
    fields = re.findall(r"\b\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", body)
    assert fields == [f[0] for f in _capi.GemmDesc._fields_], fields


def test_argument_errors_are_reported_without_a_gpu():
    """argument validation happens before any launch, so it can be exercised on a CPU-only box"""
    from sam_textvqa_amd import _capi
    with pytest.raises(_capi.SamHipError) as e:
        _capi.call("sam_attn_fwd", None, None, 0, 0, 1, 8, 12, 32, 0.1, 0.0, 0, 0, None, None, None, None)
    assert "head_dim" in str(e.value)
    d = _capi.GemmDesc()
    with pytest.raises(_capi.SamHipError):
        _capi.call("sam_gemm_bf16", d, None)          # empty problem


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_path_fails_loudly_without_gpu():
    import sam_textvqa_amd.modules as M
    from sam_textvqa_amd import ops
    from sam_textvqa_amd._capi import SamHipError
    with pytest.raises(SamHipError):
        ops.mask_bits_prefix_lm(torch.ones(2, 8, dtype=torch.uint8), 2)       # CPU tensor: rejected, no fallback
    cfg = M.BertConfig.from_dict(dict(hidden_size=768, num_spatial_relations=12, max_seq_length=4, num_decoding_steps=2,
                                      attention_mask_quadrants=[1, 2], intermediate_size=64))
    layer = M.SpatialBertLayer(cfg)
    with pytest.raises(Exception):
        layer(torch.zeros(1, 10, 768), torch.zeros(1, 1, 10, 10), torch.zeros(1, 4, 4, 12, dtype=torch.int8))


def test_no_oracle_import_in_product_package():
    pkg = os.path.join(ROOT, "sam-textvqa_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, "%s imports the oracle" % f


def test_plain_cxx_host_program_links_against_the_c_abi(tmp_path):
    """tests/cabi/host_smoke.cpp includes only sam_hip.h and the HIP runtime: it must compile and link (it runs in the gpu suite)"""
    from tests.cabi_host import build_host_smoke
    exe = build_host_smoke(tmp_path)
    assert os.path.exists(exe)


def test_a_library_built_from_other_sources_is_refused(tmp_path, monkeypatch):
    """_capi.lib() compares the digest compiled into the binary with the tree's: a stale libsam_hip.so never runs against changed signatures"""
    import sam_textvqa_amd._build as b
    from sam_textvqa_amd import _capi
    _capi.lib()
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(b, "build", lambda *a, **k: b.LIB)                # pretend the rebuild was skipped ...
    monkeypatch.setattr(b, "_digest", lambda: "0" * 64)                  # ... although the sources changed
    with pytest.raises(_capi.SamHipError, match="other sources"):
        _capi.lib()
    monkeypatch.setattr(b, "build", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hipcc failed on gemm.hip")))
    with pytest.raises(_capi.SamHipError, match="could not be"):
        _capi.lib()                                                        # a failed rebuild is an error even though an older .so exists


def test_torch_custom_ops_build_and_register_without_a_gpu():
    """csrc_torch/sam_torch_ops.cpp: TORCH_LIBRARY(sam_hip) schemas are visible to the dispatcher after load; running them needs the GPU"""
    from sam_textvqa_amd import torchops
    ns = torchops.ns()
    for op in ("linear", "spatial_attn_fwd", "spatial_attn_bwd", "spatial_attn_fwd_train", "spatial_attn_bwd_fused", "layernorm_fwd", "layernorm_bwd", "encoder_layer_fwd",
               "encoder_layer_bwd"):
        assert hasattr(ns, op), op
    schema = str(torch.ops.sam_hip.encoder_layer_fwd.default._schema)
    assert "Tensor[] params" in schema and "int[] seeds" in schema
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, NotImplementedError)):
            torch.ops.sam_hip.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-12)       # no CPU kernel is registered
