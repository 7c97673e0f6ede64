"""CPU: the C-ABI library loads and exports every symbol include/sfft_amd.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "sfft_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sfft_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from sfft_amd.build import build_library
    build_library()
    from sfft_amd import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert b"gfx950" in lib.sfft_version()


def test_header_enums_match_python_binding():
    from sfft_amd import _lib
    txt = open(os.path.join(ROOT, "include", "sfft_amd.h")).read()
    q = re.search(r"SFFT_Q_N0 = 0,(.*?)SFFT_Q_COUNT", txt, re.S).group(1)
    fields = ["N0"] + [f.strip()[len("SFFT_Q_"):] for f in re.sub(r"/\*.*?\*/", "", q, flags=re.S).split(",") if f.strip()]
    assert [f.upper() for f in _lib.QUERY_FIELDS] == [f.upper().replace("CONSTPHOTRATIO", "CONSTPHOTRATIO").replace("NEQ_FSFREE", "NEQ_FSFREE") for f in fields]
    st = re.search(r"SFFT_ST_PRELIM_SOLVE = 0,(.*?)SFFT_ST_COUNT", txt, re.S).group(1)
    stages = ["PRELIM_SOLVE"] + [f.strip()[len("SFFT_ST_"):] for f in re.sub(r"/\*.*?\*/", "", st, flags=re.S).split(",") if f.strip()]
    assert [s.upper() for s in _lib.STAGES] == stages, (stages, _lib.STAGES)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sfft_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(_lib.SfftLibraryMissing, match="no CPU fallback"):
        _lib.lib()


def test_ssc_argument_errors_need_no_gpu():
    from sfft_amd.sfftcore import SingleSFFTConfigure
    with pytest.raises(Exception, match="KerPolyOrder should be 0/1/2/3"):
        SingleSFFTConfigure.SSC(64, 64, 2, KerPolyOrder=4, VERBOSE_LEVEL=0)
    with pytest.raises(Exception, match="BGPolyOrder should be 0/1/2/3"):
        SingleSFFTConfigure.SSC(64, 64, 2, BGPolyOrder=5, VERBOSE_LEVEL=0)
    with pytest.raises(Exception, match="dramatically small size"):
        SingleSFFTConfigure.SSC(4, 64, 2, VERBOSE_LEVEL=0)
    with pytest.raises(Exception, match="no CPU path"):
        SingleSFFTConfigure.SSC(64, 64, 2, BACKEND_4SUBTRACT="Numpy", VERBOSE_LEVEL=0)


def test_minifits_roundtrip(tmp_path):
    import numpy as np
    from sfft_amd.utils import minifits
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64, np.int16):
        a = (rng.normal(size=(7, 11)) * 100).astype(dt)
        path = str(tmp_path / ("t_%s.fits" % np.dtype(dt).name))
        cards = []
        minifits.set_card(cards, "KERHW", 8, "MeLOn: SFFT")
        minifits.set_card(cards, "CONVD", "REF", "MeLOn: SFFT")
        minifits.writeto(path, a, cards)
        b, c2 = minifits.getdata(path)
        assert b.dtype == np.dtype(dt) and np.array_equal(a, b)
        h = minifits.header_dict(c2)
        assert h["KERHW"] == 8 and h["CONVD"] == "REF"
        assert os.path.getsize(path) % 2880 == 0
