"""Minimal FITS primary-HDU image reader / writer (numpy only).

The reference packets use astropy.io.fits (sfft/CustomizedPacket.py:4, 93-96, 191-221).  astropy is not part
of the target image, so the file-based operator falls back to this module: single primary HDU, BITPIX in
{8, 16, 32, 64, -32, -64}, BSCALE/BZERO honoured on read, header cards kept verbatim for write-back.
"""
import numpy as np

_BLOCK = 2880
_DTYPES = {8: ">u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}


def _parse_value(raw):
    v = raw.split("/")[0].strip() if not raw.strip().startswith("'") else raw.strip()
    if v.startswith("'"):
        end = v.find("'", 1)
        while end != -1 and v[end:end + 2] == "''":
            end = v.find("'", end + 2)
        return v[1:end].rstrip() if end != -1 else v[1:].rstrip()
    if v in ("T", "F"):
        return v == "T"
    try:
        return int(v)
    except ValueError:
        try:
            return float(v.replace("D", "E"))
        except ValueError:
            return v


def read_header(buf, offset=0):
    cards, pos, done = [], offset, False
    while not done:
        block = buf[pos:pos + _BLOCK]
        if len(block) < _BLOCK:
            raise IOError("truncated FITS header")
        for k in range(0, _BLOCK, 80):
            card = block[k:k + 80].decode("ascii", "replace")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            cards.append(card)
        pos += _BLOCK
    return cards, pos


def header_dict(cards):
    d = {}
    for c in cards:
        if c[8:10] == "= ":
            d[c[:8].strip()] = _parse_value(c[10:])
    return d


def getdata(path):
    """Return (data, cards): data indexed [NAXIS2][NAXIS1] like astropy's fits.getdata, native byte order."""
    with open(path, "rb") as f:
        buf = f.read()
    cards, pos = read_header(buf)
    h = header_dict(cards)
    if not h.get("SIMPLE", False):
        raise IOError("%s: not a simple FITS file" % path)
    naxis = int(h.get("NAXIS", 0))
    if naxis != 2:
        raise IOError("%s: primary HDU must be a 2-D image (NAXIS=%d)" % (path, naxis))
    n1, n2, bitpix = int(h["NAXIS1"]), int(h["NAXIS2"]), int(h["BITPIX"])
    dt = np.dtype(_DTYPES[bitpix])
    raw = np.frombuffer(buf, dtype=dt, count=n1 * n2, offset=pos).reshape(n2, n1)
    bscale, bzero = float(h.get("BSCALE", 1.0)), float(h.get("BZERO", 0.0))
    if bscale != 1.0 or bzero != 0.0:
        data = raw.astype(np.float64) * bscale + bzero
    else:
        data = raw.astype(dt.newbyteorder("="))
    return data, cards


def _card(key, value, comment=""):
    if isinstance(value, bool):
        v = "%20s" % ("T" if value else "F")
    elif isinstance(value, (int, np.integer)):
        v = "%20d" % value
    elif isinstance(value, (float, np.floating)):
        v = "%20s" % repr(float(value)).upper()
    else:
        s = "'%-8s'" % str(value).replace("'", "''")
        v = "%-20s" % s
    c = "%-8s= %s" % (key[:8], v)
    if comment:
        c += " / " + comment
    return c[:80].ljust(80)


def set_card(cards, key, value, comment=""):
    new = _card(key, value, comment)
    for k, c in enumerate(cards):
        if c[:8].strip() == key:
            cards[k] = new
            return
    cards.append(new)


def writeto(path, data, cards=None):
    """Write a 2-D image as the primary HDU.  `cards` (from getdata) are kept except the structural ones."""
    data = np.asarray(data)
    bitpix = {np.dtype("u1"): 8, np.dtype("i2"): 16, np.dtype("i4"): 32, np.dtype("i8"): 64,
              np.dtype("f4"): -32, np.dtype("f8"): -64}.get(data.dtype.newbyteorder("="))
    if bitpix is None:
        data = data.astype(np.float64)
        bitpix = -64
    n2, n1 = data.shape
    out = [_card("SIMPLE", True, "conforms to FITS standard"), _card("BITPIX", bitpix, "array data type"),
           _card("NAXIS", 2, "number of array dimensions"), _card("NAXIS1", n1), _card("NAXIS2", n2)]
    skip = {"SIMPLE", "BITPIX", "NAXIS", "NAXIS1", "NAXIS2", "BSCALE", "BZERO", "EXTEND", "END"}
    for c in (cards or []):
        if c[:8].strip() not in skip:
            out.append(c[:80].ljust(80))
    out.append("END".ljust(80))
    hdr = "".join(out)
    hdr += " " * ((-len(hdr)) % _BLOCK)
    payload = data.astype(np.dtype(_DTYPES[bitpix])).tobytes()
    payload += b"\0" * ((-len(payload)) % _BLOCK)
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(payload)
