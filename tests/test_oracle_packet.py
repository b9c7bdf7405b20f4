"""CPU: bftkv wire-packet framing restatement (packet/packet.go) -- round trips and the optional-field rules."""
import struct

import pytest

from corpus import build as cb
from oracle import packet as pk


def test_serialize_layout_matches_reference_framing():
    # packet.go:35-60: chunk = u64be len + bytes; t = u64be; nil sig = 22 zero bytes (packet.go:192-212)
    b = pk.serialize(b"x", b"vv", 7)
    assert b == struct.pack(">Q", 1) + b"x" + struct.pack(">Q", 2) + b"vv" + struct.pack(">Q", 7)
    assert pk.serialize(b"x", b"v", 1, None) == pk.serialize(b"x", b"v", 1) + bytes(22)
    s = pk.SignaturePacket(Type=1, Version=3, Completed=True, Data=b"DD", Cert=b"C")
    w = pk.write_signature(s)
    assert w == b"\x01" + struct.pack(">I", 3) + b"\x01" + struct.pack(">Q", 2) + b"DD" + struct.pack(">Q", 1) + b"C"
    # the independent generator agrees byte for byte
    assert cb.serialize_tbs(b"x", b"vv", 7) == b
    assert cb.sigpkt(b"DD", b"C", completed=True) == b"\x01" + struct.pack(">I", 0) + b"\x01" + w[6:]


def test_parse_round_trip_and_optional_fields():
    sig = pk.SignaturePacket(1, 0, False, b"sigdata", b"cert")
    ss = pk.SignaturePacket(1, 0, True, b"ssdata", None)
    full = pk.serialize(b"var", b"val", 42, sig, ss, b"auth")
    x, v, t, s1, s2, a = pk.parse(full)
    assert (x, v, t, a) == (b"var", b"val", 42, b"auth")
    assert (s1.Data, s1.Cert, s1.Completed) == (b"sigdata", b"cert", False)
    assert (s2.Data, s2.Cert, s2.Completed) == (b"ssdata", None, True)
    # trailing fields may be absent (io.EOF => nil), packet.go:62-115
    assert pk.parse(pk.serialize(b"var"))[:3] == (b"var", None, 0)
    assert pk.parse(pk.serialize(b"var", b"val", 42))[3:] == (None, None, None)
    assert pk.parse(pk.serialize(b"var", b"val", 42, sig))[4] is None
    # Type 0 reads back as nil (packet.go:231-233); empty chunk reads back as nil (:126-133)
    assert pk.parse(pk.serialize(b"var", b"", 1, None, ss))[1] is None
    assert pk.parse(pk.serialize(b"var", b"val", 1, None, ss))[3] is None
    with pytest.raises(pk.PacketError):
        pk.parse(b"")
    with pytest.raises(pk.PacketError):
        pk.parse(full[:-2])   # short read inside the last chunk: io.ErrUnexpectedEOF


def test_tbs_and_tbss_are_prefixes():
    sig = pk.SignaturePacket(1, 0, False, b"s" * 287, b"c" * 900)
    ss = pk.SignaturePacket(1, 0, False, b"t" * 600, None)
    req = pk.serialize(b"key00000001", b"v" * 64, 9, sig, ss)
    assert pk.tbs(req) == pk.serialize(b"key00000001", b"v" * 64, 9)
    assert pk.tbss(req) == pk.serialize(b"key00000001", b"v" * 64, 9, sig)
    assert req.startswith(pk.tbss(req)) and pk.tbss(req).startswith(pk.tbs(req))
    assert pk.parse_signature(pk.serialize_signature(sig)).Data == sig.Data
    with pytest.raises(pk.PacketError):
        pk.tbss(pk.serialize(b"a", b"b", 1))   # no signature to skip
