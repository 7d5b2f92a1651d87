"""Two things a caller outside the reference's own world needs from the drop-in:
  * an Adler-32 that standard zlib accepts (the reference's signed-remainder value, the default,
    differs from RFC 1950's on bulk data above 0x7F): crc_op ZIPC_HIP_CRC_ADLER32_RFC1950 and
    zipc_hip_set_adler_rfc1950();
  * calls from several threads at once (the reference module is re-entrant): the Python mirror
    and the C++ host layer give every thread its own context."""
import threading
import zlib

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

HIGH = [b"\xff" * 20000, util.rand_bytes(100000, 77), bytes(range(128, 256)) * 900, util.rand_bytes(6000, 5)]


def test_default_adler_is_the_references_and_zlib_rejects_it(gpu_ctx, oracle):
    from zipc_amd import zipc_deflate as Z

    for d in HIGH:
        adler, comp = Z.zlib_compress(d, level="default").get_ok()
        assert adler == oracle.zlib_compress(d, level=2)[2] and adler != zlib.adler32(d)
        with pytest.raises(zlib.error):
            zlib.decompress(comp)
        assert Z.zlib_decompress(zlib.compress(d), decompressed_size=len(d)).is_error()  # as the reference would
        assert Z.zlib_decompress(comp, decompressed_size=len(d)).get_ok()[0] == d        # its own trailers it accepts


def test_rfc1950_mode_interoperates_with_zlib():
    import torch

    import zipc_amd
    from zipc_amd import batch, zipc_deflate as Z

    ctx = zipc_amd.Context(0)
    ctx.set_adler_rfc1950(True)
    for d in HIGH + [b"", b"a", util.text(50000, 3)]:
        for level in ("none", "fast", "default", "best"):
            adler, comp = Z.zlib_compress(d, level=level, ctx=ctx).get_ok()
            assert adler == zlib.adler32(d)
            assert zlib.decompress(comp) == d
        back, adler = Z.zlib_decompress(zlib.compress(d, 6), decompressed_size=len(d), ctx=ctx).get_ok()
        assert back == d and adler == zlib.adler32(d)
    dev = torch.device("cuda", 0)
    buf = torch.from_numpy(np.frombuffer(util.rand_bytes(3 * 5552 * 1000 + 17, 9), np.uint8).copy()).to(dev)
    _, adler = batch.checksum_device(ctx, buf, want_crc32=False)
    assert adler == zlib.adler32(buf.cpu().numpy().tobytes())
    # batch forms: crc_op 3 per stream, deflate side and inflate side
    streams = HIGH + [util.text(70000, 1), util.rand_bytes(200000, 3)]
    n = len(streams)
    src_off = np.cumsum([0] + [len(s) for s in streams[:-1]]).astype(np.uint64)
    caps = [batch.deflate_bound(len(s)) for s in streams]
    slots = [(c + 255) // 256 * 256 for c in caps]
    dst_off = np.cumsum([0] + slots[:-1]).astype(np.uint64)
    descs = batch.make_descs(src_off, [len(s) for s in streams], dst_off, caps)
    src = torch.from_numpy(np.frombuffer(b"".join(streams) + b"\0" * 64, dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(int(sum(slots)) + 256, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.deflate_batch(ctx, src, dst, batch.to_device(descs, dev), d_res, n, max(map(len, streams)), sum(map(len, streams)), 2, 3)
    res = batch.results_from_device(d_res)
    assert [int(c) for c in res["checksum"]] == [zlib.adler32(s) for s in streams]
    lim = max(map(len, streams))
    idescs = batch.make_descs(dst_off, res["out_len"], np.arange(n, dtype=np.uint64) * (lim + 256), [lim] * n, limit=[len(s) for s in streams])
    out = torch.zeros(n * (lim + 256) + 256, dtype=torch.uint8, device=dev)
    d_ires = torch.zeros(n * 16, dtype=torch.uint8, device=dev)
    batch.inflate_batch(ctx, dst, out, batch.to_device(idescs, dev), d_ires, n, lim, 3)
    ires = batch.results_from_device(d_ires)
    assert (ires["status"] == 0).all() and [int(c) for c in ires["checksum"]] == [zlib.adler32(s) for s in streams]
    ctx.close()


def test_two_threads_round_trip_at_once(oracle):
    from zipc_amd import zipc_deflate as Z
    from zipc_amd import zipc_host

    errors = []

    def python_mirror(seed):
        try:
            for k in range(12):
                d = util.rand_bytes(200000 + 977 * k, seed * 100 + k, 4) + util.text(30000, seed + k)
                crc, comp = Z.crc_32_and_deflate(d, level="default").get_ok()
                assert comp == oracle.deflate(d, level=2)[1] and crc == zlib.crc32(d)
                assert Z.inflate(comp, decompressed_size=len(d)).get_ok() == d
        except Exception as e:  # noqa: BLE001
            errors.append(("mirror", seed, repr(e)))

    def host_layer(seed):
        try:
            for k in range(4):
                files = [(b"f/%d_%d.bin" % (seed, i), util.rand_bytes(50000 + 31 * i, seed * 1000 + 10 * k + i, 3)) for i in range(8)]
                a = zipc_host.Archive()
                a.add_files_deflate(files, level=2)
                got = dict(zipc_host.Archive.of_binary_string(a.to_binary_string()).extract_all())
                assert all(got[p] == d for p, d in files)
        except Exception as e:  # noqa: BLE001
            errors.append(("host", seed, repr(e)))

    threads = [threading.Thread(target=python_mirror, args=(s,)) for s in (1, 2)]
    threads += [threading.Thread(target=host_layer, args=(s,)) for s in (3, 4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []
