"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads,
exports every symbol include/psamd.h declares, and refuses to run without a GPU
(no fallback path exists)."""
import ctypes as C
import os
import re

import pytest

import particlesystem_amd as ps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    ps.build()
    return ps.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "psamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psamd_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 30
    bound = {n for n, _, _ in ps.ABI}
    for n in names:
        assert hasattr(lib, n), "libpsamd.so does not export " + n
        assert n in bound, "python mirror does not bind " + n


def test_abi_version_and_defaults(lib):
    # the header, the library and the python mirror agree on the layout version
    header = int(re.search(r"#define PSAMD_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "psamd.h")).read()).group(1))
    assert lib.psamd_abi_version() == header == ps.ABI_VERSION
    cfg = ps.default_config()
    assert (cfg.max_particles_num, cfg.x_factor, cfg.chunk_factor, cfg.chunk_dim) == (1 << 20, 2, 4, 4)
    assert (cfg.cell_size, cfg.eps2, cfg.collision_radius, cfg.dt) == (5.0, 0.2, 0.4, 0.05)
    assert lib.psamd_status_string(2).decode().startswith("no usable HIP device")


def test_a_library_of_another_abi_version_is_refused(tmp_path, monkeypatch):
    """PSAMD_LIB may point at another build (A/B runs): struct layouts of another version must not be bound."""
    monkeypatch.setattr(ps, "ABI_VERSION", ps.ABI_VERSION + 1)
    monkeypatch.setattr(ps, "_lib", None)
    with pytest.raises(RuntimeError, match="ABI version"):
        ps.load()
    monkeypatch.undo()
    ps._lib = None
    ps.load()


def test_null_arguments_are_rejected(lib):
    assert lib.psamd_default_config(None) == 1
    assert lib.psamd_create(None, None) == 1
    assert lib.psamd_step(None, 1) == 1
    assert lib.psamd_destroy(None) == 0
    assert lib.psamd_set_graphs(None, 1) == 1 and lib.psamd_set_wait_policy(None, 1) == 1
    assert lib.psamd_get_graph_stats(None, None, None) == 1


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(ps.PsamdError) as e:
        ps.ParticleSystem(ps.default_config())
    assert e.value.status in (2, 3)     # PSAMD_ERR_NO_DEVICE (or a HIP init error), never a silent fallback


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "particlesystem_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower() or f == "__init__.py" and "oracle" not in src.lower(), \
                    f + " mentions the oracle"
