"""Every query lookup layout returns the oracle's results: the hashed {component id, weight} entries (one LDS
read per document component; the default for u32 components) forced onto the fuzz seeds - u16 and u32
components, f16 and fixed-u8 values, 512- and 1024-thread workgroups, cooperative and plain launches."""
import pytest

from test_gpu_fuzz import test_differential as _differential

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1, 4, 7, 9, 14, 15, 16, 18, 19, 20, 22, 24])
def test_hashed_entries_on_the_fuzz_seeds(seed, monkeypatch):
    monkeypatch.setenv("SGPU_FORCE_HASH", "1")
    if seed % 3 == 0:
        monkeypatch.setenv("SGPU_COOP", "force")
    _differential(seed, monkeypatch)


@pytest.mark.parametrize("seed", [2, 9, 19])
def test_without_the_hashed_entries(seed, monkeypatch):
    monkeypatch.setenv("SGPU_NO_HASH", "1")
    _differential(seed, monkeypatch)
