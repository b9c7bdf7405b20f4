"""CPU: the HTTP framing half of the wire replay (bftkv_amd/wire.py) -- message parsing, the router of
transport/http/http.go:104-141, request/response pairing, and the verdict table -- without a GPU (the signatures are the
-m gpu test's business, tests/test_gpu_protocol.py::test_http_wire_replay)."""
import pytest

from bftkv_amd import wire


def _req(path, body=b"", method="POST", chunked=False):
    if chunked:
        half = len(body) // 2
        b = b"%x\r\n%s\r\n%x;ext=1\r\n%s\r\n0\r\nX-trailer: t\r\n\r\n" % (half, body[:half], len(body) - half, body[half:])
        return b"%s %s HTTP/1.1\r\nHost: a01\r\nTransfer-Encoding: chunked\r\n\r\n" % (method.encode(), path.encode()) + b
    return b"%s %s HTTP/1.1\r\nHost: a01\r\nContent-Type: application/octet-stream\r\nContent-Length: %d\r\n\r\n" % (method.encode(), path.encode(), len(body)) + body


def _resp(status, body=b"", xerr=None):
    h = b"HTTP/1.1 %d X\r\n" % status
    if xerr is not None:
        h += b"X-Error: %s\r\n" % xerr.encode()          # net/http canonicalises the name; Get is case-insensitive
    return h + b"Content-Length: %d\r\n\r\n" % len(body) + body


def test_router_follows_serve_http():
    assert wire.command_of("/bftkv/v1/write") == "write"
    assert wire.command_of("/BFTKV/V1/Write") == "write"                 # strings.ToLower(r.URL.Path)
    assert wire.command_of("/bftkv/v1/write?x=1") == "write"             # URL.Path excludes the query
    assert wire.command_of("http://a01:5701/bftkv/v1/sign") == "sign"    # absolute-form target
    assert wire.command_of("/bftkv/v1/write/") is None and wire.command_of("/bftkv/v1/") is None
    assert wire.command_of("/bftkv/v2/write") is None and wire.command_of("/x/bftkv/v1/write") is None
    assert wire.command_of("/bftkv/v1/writes") is None
    assert [wire.command_of("/bftkv/v1/" + c) for c in wire.COMMANDS] == list(wire.COMMANDS) and len(wire.COMMANDS) == 13


def test_http_stream_parsing_and_pairing():
    body = bytes(range(256)) * 5
    cap = (_req("/bftkv/v1/write", body) + _resp(200, b"reply") + b"\r\n" +
           _req("/bftkv/v1/sign", body, chunked=True) + _resp(500, b"Internal Server Error\n", "crypto: invalid signature") +
           _req("/nothing", b"zz") + _resp(404, b"404 page not found\n") +
           _req("/bftkv/v1/time", b""))                                   # logged request without an answer
    msgs = wire.parse_http_stream(cap)
    assert [m.is_request for m in msgs] == [True, False, True, False, True, False, True]
    assert msgs[0].body == body and msgs[2].body == body and msgs[1].body == b"reply" and msgs[6].body == b""
    assert msgs[3].status == 500 and msgs[3].header("x-error") == "crypto: invalid signature" and msgs[1].header("X-error") == ""
    exs = wire.pair_exchanges(msgs)
    assert [(e.command, e.response.status if e.response else None) for e in exs] == [("write", 200), ("sign", 500), (None, 404), ("time", None)]
    for bad in (cap[:-1] + b"", b"POST /x HTTP/1.1\r\nContent-Length: 10\r\n\r\nshort", b"GARBAGE\r\n\r\n", b"HTTP/1.1 200 OK\r\nContent-Length: 0\r\n\r\n"):
        if bad is cap[:-1]:
            continue
        with pytest.raises(wire.CaptureError):
            wire.pair_exchanges(wire.parse_http_stream(bad))


def test_verdict_table():
    mk = lambda cmd, transport, site_error, resp: wire.Exchange(0, cmd, wire.parse_http_stream(_req("/bftkv/v1/" + (cmd or "x")))[0],
                                                                 wire.parse_http_stream(resp)[0] if resp else None, transport=transport, site_error=site_error)
    J = wire._judge
    assert J(mk("write", "ok", None, _resp(200))) == "consistent"
    assert J(mk("write", "ok", None, _resp(500, xerr="bad timestamp"))) == "not-judged"            # storage-dependent: outside this path
    assert J(mk("write", "ok", wire.ERR_INSUFFICIENT_SIGNATURES, _resp(500, xerr=wire.ERR_INSUFFICIENT_SIGNATURES))) == "consistent"
    assert J(mk("write", "ok", wire.ERR_INSUFFICIENT_SIGNATURES, _resp(200))) == "accepted-but-fails-verification"
    assert J(mk("write", "ok", wire.ERR_INSUFFICIENT_SIGNATURES, _resp(500, xerr=wire.ERR_MALFORMED))) == "wrong-error"
    assert J(mk("sign", "unverified", wire.ERR_CERT_NOT_FOUND, _resp(500, xerr="crypto: certifiate not found"))) == "consistent"
    assert J(mk("sign", wire.ERR_DECRYPTION_FAILED, None, _resp(500, xerr=wire.ERR_DECRYPTION_FAILED))) == "consistent"
    assert J(mk("sign", wire.ERR_DECRYPTION_FAILED, None, _resp(200))) == "accepted-but-fails-verification"
    assert J(mk("read", wire.ERR_INVALID_SIGNATURE, None, _resp(500, xerr="openpgp: invalid signature: hash tag doesn't match"))) == "consistent"
    assert J(mk("read", "fenced", None, _resp(200))) == "not-judged" and J(mk("write", "ok", "fenced", _resp(200))) == "not-judged"
    assert J(mk("write", "ok", None, None)) == "no-response" and J(mk(None, "", None, _resp(404))) == "consistent"
    assert J(mk(None, "", None, _resp(200))) == "routed-but-unknown-path"
