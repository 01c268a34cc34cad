"""A pure-Python MP4 path for the CLI's video input / output: Motion-JPEG in an ISO base-media (.mp4)
container, PIL for the frames, nothing else.

The reference reads videos through torchvision.io / imageio and writes ``masked_in.mp4`` / ``inpaint_out.mp4`` with
``imageio.mimwrite(..., fps=fps, quality=7)`` (inference_propainter.py:49-67,471-472) -- both need ffmpeg, which this image does not
have.  H.264 cannot be encoded or decoded without it, but an .mp4 FILE does not have to be H.264: ISO/IEC 14496-1 registers JPEG as
object type 0x6C of the ``mp4v`` sample entry, which ffmpeg, VLC and browsers' fallbacks read.  So:

  * ``write_mp4(path, frames, fps, quality)``: every frame JPEG-encoded by PIL, muxed as one video track (ftyp / mdat / moov with
    mvhd, tkhd, mdhd, hdlr, vmhd, dref, stsd[mp4v + esds 0x6C], stts, stsc, stsz, stco / co64).  ``quality`` follows imageio's 0..10
    scale (7 -> JPEG quality 85), so the CLI passes the reference's value through.
  * ``read_mp4(path)``: parses the boxes of ANY ISO-BMFF / QuickTime file, and decodes the frames of tracks whose samples are JPEG
    (``mp4v`` with object type 0x6C, ``jpeg``, ``mjpa``, ``mjpb`` is refused) -- files written by ``write_mp4`` or by
    ``ffmpeg -c:v mjpeg``.  Any other codec (``avc1``, ``hvc1``, ...) raises ``UnsupportedCodec`` naming it: those need
    imageio + ffmpeg, which ``video_io`` tries first when they are installed.

Frames are intra-only, so the round trip is exact up to JPEG quantisation (tests/test_cli_cpu.py: 33 dB at quality 7 = JPEG 85 on
noisy synthetic frames, the frame count and the frame rate are preserved)."""
import io
import struct

import numpy as np
from PIL import Image

JPEG_SAMPLE_ENTRIES = (b"jpeg", b"mjpa", b"MJPG", b"mjpg")


class UnsupportedCodec(RuntimeError):
    pass


def _box(kind, *payload):
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), kind) + body


def _full(kind, version, flags, *payload):
    return _box(kind, struct.pack(">I", (version << 24) | flags), *payload)


def jpeg_quality(imageio_quality):
    """imageio's 0..10 video quality -> PIL JPEG quality (7 -> 85; clamped to 30..95)."""
    q = 5.0 if imageio_quality is None else float(imageio_quality)
    return int(min(95, max(30, round(50 + 5 * q))))


def write_mp4(path, frames, fps=24.0, quality=7):
    """frames: iterable of uint8 [H,W,3] RGB arrays (same size).  Returns the number of frames written."""
    frames = [np.ascontiguousarray(np.asarray(f)[..., :3], dtype=np.uint8) for f in frames]
    if not frames:
        raise ValueError("write_mp4: no frames")
    h, w = frames[0].shape[:2]
    if any(f.shape[:2] != (h, w) for f in frames):
        raise ValueError("write_mp4: all frames must have the same size")
    q = jpeg_quality(quality)
    samples = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=q, subsampling=0 if q >= 85 else 2)
        samples.append(buf.getvalue())
    n = len(samples)
    fps = float(fps or 24.0)
    timescale = 90000
    delta = max(1, int(round(timescale / fps)))
    duration = delta * n
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2mp41")
    mdat_payload = b"".join(samples)
    big = len(mdat_payload) + 16 >= (1 << 32)
    mdat = (struct.pack(">I4sQ", 1, b"mdat", 16 + len(mdat_payload)) if big else struct.pack(">I4s", 8 + len(mdat_payload), b"mdat")) + mdat_payload
    first = len(ftyp) + (16 if big else 8)
    offsets, o = [], first
    for s in samples:
        offsets.append(o)
        o += len(s)
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration), struct.pack(">IH", 0x10000, 0x100), b"\0" * 10, matrix,
                 b"\0" * 24, struct.pack(">I", 2))
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration), b"\0" * 8, struct.pack(">HHHH", 0, 0, 0, 0), matrix,
                 struct.pack(">II", w << 16, h << 16))
    mdhd = _full(b"mdhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration), struct.pack(">HH", 0x55C4, 0))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide"), b"\0" * 12, b"VideoHandler\0")
    vmhd = _full(b"vmhd", 0, 1, b"\0" * 8)
    dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1)))
    # esds: ES_Descriptor(tag 3){ES_ID 1, flags 0, DecoderConfigDescriptor(tag 4){objectType 0x6C = JPEG, streamType visual, buffer size,
    # max / avg bitrate}, SLConfigDescriptor(tag 6){predefined 2}}
    biggest = max(len(s) for s in samples)
    bitrate = int(len(mdat_payload) * 8 * fps / n)
    dcd = struct.pack(">BB", 0x6C, 0x11) + struct.pack(">I", biggest)[1:] + struct.pack(">II", bitrate, bitrate)
    esd = struct.pack(">HB", 1, 0) + bytes([4, len(dcd)]) + dcd + bytes([6, 1, 2])
    esds = _full(b"esds", 0, 0, bytes([3, len(esd)]) + esd)
    entry = _box(b"mp4v", b"\0" * 6, struct.pack(">H", 1), b"\0" * 16, struct.pack(">HH", w, h), struct.pack(">II", 0x480000, 0x480000),
                 struct.pack(">I", 0), struct.pack(">H", 1), b"\0" * 32, struct.pack(">Hh", 24, -1), esds)
    stsd = _full(b"stsd", 0, 0, struct.pack(">I", 1), entry)
    stts = _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta))
    stsc = _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1))
    stsz = _full(b"stsz", 0, 0, struct.pack(">II", 0, n), b"".join(struct.pack(">I", len(s)) for s in samples))
    if offsets[-1] >= (1 << 32):
        stco = _full(b"co64", 0, 0, struct.pack(">I", n), b"".join(struct.pack(">Q", x) for x in offsets))
    else:
        stco = _full(b"stco", 0, 0, struct.pack(">I", n), b"".join(struct.pack(">I", x) for x in offsets))
    stbl = _box(b"stbl", stsd, stts, stsc, stsz, stco)
    moov = _box(b"moov", mvhd, _box(b"trak", tkhd, _box(b"mdia", mdhd, hdlr, _box(b"minf", vmhd, dinf, stbl))))
    with open(path, "wb") as f:
        f.write(ftyp)
        f.write(mdat)
        f.write(moov)
    return n


# ------------------------------------------------------------------------------------------------------------------ reader
CONTAINERS = (b"moov", b"trak", b"mdia", b"minf", b"stbl", b"dinf", b"edts", b"udta")


def _boxes(data, lo, hi):
    """Yields (kind, payload_lo, payload_hi) of the boxes in data[lo:hi]."""
    p = lo
    while p + 8 <= hi:
        size, kind = struct.unpack_from(">I4s", data, p)
        head = 8
        if size == 1:
            size = struct.unpack_from(">Q", data, p + 8)[0]
            head = 16
        elif size == 0:
            size = hi - p
        if size < head or p + size > hi:
            raise ValueError(f"corrupt box {kind!r} at {p}: size {size}")
        yield kind, p + head, p + size
        p += size


def _find(data, lo, hi, path):
    for kind, a, b in _boxes(data, lo, hi):
        if kind == path[0]:
            if len(path) == 1:
                yield a, b
            else:
                yield from _find(data, a, b, path[1:])


def _esds_object_type(data, a, b):
    """objectTypeIndication of the DecoderConfigDescriptor inside an esds box payload (after version / flags), or None."""
    p = a + 4

    def desc(p):
        tag = data[p]
        p += 1
        size = 0
        for _ in range(4):
            c = data[p]
            p += 1
            size = (size << 7) | (c & 0x7F)
            if not c & 0x80:
                break
        return tag, p, size
    try:
        tag, p, _ = desc(p)
        if tag != 3:
            return None
        flags = data[p + 2]
        p += 3 + (2 if flags & 0x80 else 0) + (2 if flags & 0x20 else 0)
        if flags & 0x40:
            p += 1 + data[p]
        tag, p, _ = desc(p)
        return data[p] if tag == 4 else None
    except IndexError:
        return None


def read_mp4(path):
    """-> (frames: list of RGB PIL images, fps: float).  JPEG-coded video tracks only (see the module docstring)."""
    with open(path, "rb") as f:
        data = f.read()
    top = {k for k, _, _ in _boxes(data, 0, len(data))}
    if b"moov" not in top:
        raise ValueError(f"{path}: no moov box -- not an MP4 / QuickTime file (or a fragmented one)")
    seen = []
    for ta, tb in _find(data, 0, len(data), (b"moov", b"trak")):
        hd = list(_find(data, ta, tb, (b"mdia", b"hdlr")))
        if not hd or data[hd[0][0] + 8:hd[0][0] + 12] != b"vide":
            continue
        (ma, mb), = list(_find(data, ta, tb, (b"mdia", b"mdhd")))[:1]
        ver = data[ma]
        timescale = struct.unpack_from(">I", data, ma + (20 if ver == 1 else 12))[0]
        (sa, sb), = list(_find(data, ta, tb, (b"mdia", b"minf", b"stbl")))[:1]
        tbl = {k: (a, b) for k, a, b in _boxes(data, sa, sb)}
        a, b = tbl[b"stsd"]
        entry_kind, ea, eb = next(_boxes(data, a + 8, b))
        codec = entry_kind
        if entry_kind == b"mp4v":
            ot = None
            for k2, a2, b2 in _boxes(data, ea + 78, eb):
                if k2 == b"esds":
                    ot = _esds_object_type(data, a2, b2)
            if ot != 0x6C:
                seen.append(f"mp4v (object type {ot:#x})" if ot is not None else "mp4v")
                continue
        elif entry_kind not in JPEG_SAMPLE_ENTRIES:
            seen.append(entry_kind.decode("latin1"))
            continue
        # sample sizes
        a, b = tbl[b"stsz"]
        uniform, n = struct.unpack_from(">II", data, a + 4)
        sizes = [uniform] * n if uniform else list(struct.unpack_from(f">{n}I", data, a + 12))
        # chunk offsets and the samples-per-chunk runs
        if b"stco" in tbl:
            a, b = tbl[b"stco"]
            nc = struct.unpack_from(">I", data, a + 4)[0]
            chunks = list(struct.unpack_from(f">{nc}I", data, a + 8))
        else:
            a, b = tbl[b"co64"]
            nc = struct.unpack_from(">I", data, a + 4)[0]
            chunks = list(struct.unpack_from(f">{nc}Q", data, a + 8))
        a, b = tbl[b"stsc"]
        ne = struct.unpack_from(">I", data, a + 4)[0]
        runs = [struct.unpack_from(">III", data, a + 8 + 12 * i) for i in range(ne)]
        offsets, si = [], 0
        for ci in range(nc):
            spc = 1
            for first, per, _ in runs:
                if ci + 1 >= first:
                    spc = per
            o = chunks[ci]
            for _ in range(spc):
                if si >= n:
                    break
                offsets.append(o)
                o += sizes[si]
                si += 1
        # frame rate from the time-to-sample table
        a, b = tbl[b"stts"]
        ne = struct.unpack_from(">I", data, a + 4)[0]
        total_d = total_n = 0
        for i in range(ne):
            cnt, d = struct.unpack_from(">II", data, a + 8 + 8 * i)
            total_d += cnt * d
            total_n += cnt
        fps = timescale * total_n / total_d if total_d else 24.0
        frames = []
        for o, sz in zip(offsets, sizes):
            chunk = data[o:o + sz]
            soi = chunk.find(b"\xff\xd8")             # (QuickTime mjpa samples carry an APP1 field header; PIL starts at SOI anyway)
            frames.append(Image.open(io.BytesIO(chunk[soi if soi > 0 else 0:])).convert("RGB"))
        return frames, float(fps)
    if seen:
        raise UnsupportedCodec(f"{path}: video codec {', '.join(seen)} -- only JPEG-coded tracks (mp4v/0x6C, jpeg, mjpa) can be decoded "
                               "without ffmpeg; install imageio + ffmpeg (video_io tries them first) or pass a folder of frames")
    raise ValueError(f"{path}: no video track")
