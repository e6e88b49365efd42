"""CPU-side checks of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every
entry point include/segearth_hip.h declares; the ctypes table matches the header; the product path
fails loudly without the extension or without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "segearth_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ("sg_create", "sg_destroy", "sg_vit_set_tensor", "sg_vit_forward", "sg_cosine_logits", "sg_stitch",
                 "sg_postprocess", "sg_adaptive_conv", "sg_outlier_suppress", "sg_similarity_map", "sg_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from clip_decontamination_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run `python -m clip_decontamination_amd.build` first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_table_matches_header():
    from clip_decontamination_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.load()
    assert lib.sg_version() >= 100
    # argument counts agree with the header prototypes
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), f"{name}: header has {n} parameters, ctypes table {len(args)}"


def test_error_channel_without_gpu():
    """Argument validation happens before any device work, so it can be exercised on CPU."""
    from clip_decontamination_amd import _lib
    lib = _lib.load()
    rc = lib.sg_cosine_logits(None, None, None, 1, 1, 1, 1, 0.0, 0.0, None, None)
    assert rc == -1
    assert b"null pointer" in lib.sg_last_error()
    with pytest.raises(RuntimeError, match="null pointer"):
        _lib.check(rc, "sg_cosine_logits")


def test_product_path_has_no_cpu_fallback():
    import torch
    from clip_decontamination_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))
    from clip_decontamination_amd.engine import HipVisionTower
    from clip_decontamination_amd import weights as Wt
    cfg = Wt.vit_config("tiny-8")
    with pytest.raises(RuntimeError, match="needs a GPU"):
        HipVisionTower(cfg, {}, "f32")


def test_struct_layouts_match_header():
    """sizeof / field order of the ctypes mirrors vs what a C compiler sees in the header."""
    import subprocess, tempfile, textwrap
    from clip_decontamination_amd import _lib
    prog = textwrap.dedent(f"""
        #include <stdio.h>
        #include <stddef.h>
        #include "{HEADER}"
        int main(void) {{
          printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(sg_vit_desc), sizeof(sg_forward_opts), sizeof(sg_tile_batch),
                 offsetof(sg_tile_batch, n_tiles), offsetof(sg_forward_opts, gem_depth), offsetof(sg_tile_batch, scene_stride));
          return 0; }}""")
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    vals = [int(v) for v in out]
    assert vals[0] == ctypes.sizeof(_lib.VitDesc)
    assert vals[1] == ctypes.sizeof(_lib.ForwardOpts)
    assert vals[2] == ctypes.sizeof(_lib.TileBatch)
    assert vals[3] == _lib.TileBatch.n_tiles.offset
    assert vals[4] == _lib.ForwardOpts.gem_depth.offset
    assert vals[5] == _lib.TileBatch.scene_stride.offset


def test_product_path_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the package, the drop-in entry points or configs/ may import it; bench.py may
    only do so inside cpu_baseline(), __graft_entry__ only inside smoke()."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node)
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append(node)
        return tree, hits

    product = glob.glob(os.path.join(root, "clip_decontamination_amd", "*.py")) + [os.path.join(root, f) for f in ("segmentor.py", "segearth_segmentor.py")]
    product += glob.glob(os.path.join(root, "configs", "*.py"))
    for path in product:
        assert not oracle_imports(path)[1], f"{path} imports the oracle"
    for path, allowed in ((os.path.join(root, "bench.py"), "cpu_baseline"), (os.path.join(root, "__graft_entry__.py"), "smoke")):
        tree, hits = oracle_imports(path)
        inside = [n for f in ast.walk(tree) if isinstance(f, ast.FunctionDef) and f.name == allowed for n in ast.walk(f)]
        for h in hits:
            assert any(h is n for n in inside), f"{path}: oracle import outside {allowed}()"
