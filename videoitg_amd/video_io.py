"""Built-in video reader for the container formats that need no codec library: MJPEG / uncompressed frames in AVI (incl. OpenDML), YUV4MPEG2, and -
round 6 - the animated image formats Pillow decodes itself (GIF, APNG, animated WebP, multi-page TIFF) and ISO base media / QuickTime files (.mp4 /
.mov) whose video track holds individually coded samples (Motion-JPEG, PNG, uncompressed RGB); an .mp4 whose track needs a codec is parsed as far as
its sample table and refused with the codec's name (`CodecNeeded`, an ImportError).

Row A2 / N3 of SURVEY.md section 8: the reference decodes with decord (``infer.py:44-51``) or PyAV (``eagle/mm_utils.py:43-103``,
``lmms_eval/models/videoitg.py:95-130``); neither is in this image, and building a codec is out of scope.  What CAN be done without one is
to read real files whose frames are individually coded: Motion-JPEG in an AVI (every frame a baseline JPEG - Pillow, already a dependency
of the image processor, decodes it), uncompressed BGR in an AVI, and raw planar YUV with a YUV4MPEG2 header.  ``VideoReader`` below exposes
the handful of calls the reference makes on ``decord.VideoReader`` (``len``, ``get_avg_fps``, ``get_batch(idx).asnumpy()``, ``vr[i].asnumpy()``,
``seek``), so ``read_video_decord`` and ``infer.py --video clip.avi`` run end to end on a real file when decord is absent.

This is host-side file parsing (struct + NumPy); nothing here runs on the GPU and nothing here is needed for the timed path.
"""
from __future__ import annotations

import io
import struct
from typing import List, Sequence, Tuple

import numpy as np

SUFFIXES = (".avi", ".y4m", ".gif", ".webp", ".apng", ".png", ".tif", ".tiff")      # (the image suffixes: ANIMATED files Pillow decodes - round 6)
# ISO base media / QuickTime files: the CONTAINER is read here, the track must be individually coded frames (Motion-JPEG, PNG, uncompressed RGB);
# a track that needs a codec (avc1, hvc1, av01, vp09, mp4v / MPEG-4 part 2 ...) raises CodecNeeded naming it
BMFF_SUFFIXES = (".mp4", ".mov", ".m4v", ".qt")


class CodecNeeded(ImportError):
    """the file parsed, its video track needs a decoder library this image does not have (an ImportError: what the reference raises without decord)"""


CODEC_NAMES = {b"avc1": "H.264", b"avc3": "H.264", b"hvc1": "HEVC", b"hev1": "HEVC", b"av01": "AV1", b"vp09": "VP9", b"vp08": "VP8", b"mp4v": "MPEG-4 part 2",
               b"s263": "H.263", b"apch": "ProRes", b"apcn": "ProRes", b"apcs": "ProRes", b"ap4h": "ProRes"}


class _Batch:
    """what decord hands back from get_batch / __getitem__: an object with .asnumpy()"""

    def __init__(self, arr: np.ndarray):
        self._arr = arr

    def asnumpy(self) -> np.ndarray:
        return self._arr


def _chunks(buf: memoryview, start: int, end: int):
    """RIFF chunks in [start, end): (fourcc, payload offset, payload size); chunks are padded to even sizes"""
    p = start
    while p + 8 <= end:
        cc = bytes(buf[p:p + 4])
        (size,) = struct.unpack_from("<I", buf, p + 4)
        yield cc, p + 8, size
        p += 8 + size + (size & 1)


class _AviIndex:
    """frame table of an AVI file (AVI 1.0, or OpenDML with 'RIFF AVIX' continuation segments): offsets / sizes of the video stream's chunks inside
    the LIST 'movi' of every segment (file order = frame order)"""

    def __init__(self, data: bytes):
        buf = memoryview(data)
        if len(buf) < 12 or bytes(buf[:4]) != b"RIFF" or bytes(buf[8:12]) != b"AVI ":
            raise ValueError("not a RIFF AVI file")
        self.frames: List[Tuple[int, int]] = []
        self.width = self.height = 0
        self.rate, self.scale, self.usec_per_frame = 0, 1, 0
        self.handler = self.compression = b""
        self.bit_count = 24
        self.top_down = False
        st = {"vid_stream": None, "n_streams": 0}
        # top level: the 'RIFF AVI ' segment, then - in OpenDML files, i.e. anything over 1 GiB - any number of 'RIFF AVIX' segments, each with
        # its own LIST 'movi' holding the following frames (file order = frame order)
        for rcc, roff, rsize in _chunks(buf, 0, len(buf)):
            if rcc != b"RIFF" or bytes(buf[roff:roff + 4]) not in (b"AVI ", b"AVIX"):
                continue
            self._segment(buf, roff + 4, min(roff + rsize, len(buf)), st)
        vid_stream = st["vid_stream"]
        if vid_stream is None or not self.width:
            raise ValueError("AVI: no video stream")

    def _segment(self, buf, start, end, st):
        """one RIFF segment's children: the header list (first segment only) and the movi list(s)"""
        for cc, off, size in _chunks(buf, start, end):
            if cc != b"LIST":
                continue
            kind = bytes(buf[off:off + 4])
            if kind == b"hdrl":
                for c2, o2, s2 in _chunks(buf, off + 4, off + size):
                    if c2 == b"avih":
                        self.usec_per_frame = struct.unpack_from("<I", buf, o2)[0]
                    elif c2 == b"LIST" and bytes(buf[o2:o2 + 4]) == b"strl":
                        is_video = False
                        for c3, o3, s3 in _chunks(buf, o2 + 4, o2 + s2):
                            if c3 == b"strh":
                                is_video = bytes(buf[o3:o3 + 4]) == b"vids" and st["vid_stream"] is None
                                if is_video:
                                    self.handler = bytes(buf[o3 + 4:o3 + 8])
                                    self.scale, self.rate = struct.unpack_from("<II", buf, o3 + 20)
                                    st["vid_stream"] = st["n_streams"]
                            elif c3 == b"strf" and is_video:
                                w, h, _planes, bits = struct.unpack_from("<iiHH", buf, o3 + 4)
                                self.compression = bytes(buf[o3 + 16:o3 + 20])
                                self.width, self.height, self.bit_count = w, abs(h), bits
                                self.top_down = h < 0
                        st["n_streams"] += 1
            elif kind == b"movi":
                if st["vid_stream"] is None:
                    raise ValueError("AVI: no video stream header before the movi list")
                tag = b"%02d" % st["vid_stream"]
                for c2, o2, s2 in _chunks(buf, off + 4, min(off + size, end)):
                    if c2[:2] == tag and c2[2:] in (b"dc", b"db"):
                        self.frames.append((o2, s2))
                    elif c2 == b"LIST" and bytes(buf[o2:o2 + 4]) == b"rec ":
                        for c3, o3, s3 in _chunks(buf, o2 + 4, o2 + s2):
                            if c3[:2] == tag and c3[2:] in (b"dc", b"db"):
                                self.frames.append((o3, s3))

    @property
    def fps(self) -> float:
        if self.rate and self.scale:
            return self.rate / self.scale
        return 1e6 / self.usec_per_frame if self.usec_per_frame else 0.0


class _BmffIndex:
    """sample table of the first video track of an ISO base media / QuickTime file (ISO/IEC 14496-12 boxes: moov > trak > mdia > minf > stbl):
    per-sample (offset, size) in decode order from stsc x stco|co64 x stsz|stz2, frame rate from mdhd's timescale and stts, the sample entry's
    fourcc / width / height / depth from stsd, and - for 'mp4v' entries - the objectTypeIndication of the esds (0x6C = JPEG, 0x6D = PNG: how
    MP4 muxers label Motion-JPEG / PNG tracks)"""

    CONTAINERS = (b"moov", b"trak", b"mdia", b"minf", b"stbl")

    def __init__(self, data: bytes):
        self.buf = buf = memoryview(data)
        self.frames: List[Tuple[int, int]] = []
        self.width = self.height = 0
        self.depth, self.codec, self.oti = 24, b"", None
        self.timescale, self.duration = 0, 0
        self._fps = 0.0
        moov = self._find(0, len(buf), b"moov")
        if moov is None:
            raise ValueError("no 'moov' box: not an ISO base media / QuickTime file (or a fragmented / truncated one)")
        try:
            for cc, s, e in self._boxes(*moov):
                if cc == b"trak" and self._track(s, e):
                    return
        except (KeyError, IndexError, struct.error) as err:          # a table the track must have is missing, or a box is shorter than its fields
            raise ValueError(f"malformed sample table in the video track ({type(err).__name__}: {err})") from err
        raise ValueError("no video track ('vide' handler) in the file")

    def _boxes(self, start: int, end: int):
        """(fourcc, payload start, box end) of the boxes in [start, end): 32-bit sizes, 64-bit 'largesize' (size 1), 'to the end' (size 0)"""
        p, buf = start, self.buf
        while p + 8 <= end:
            size, cc = struct.unpack_from(">I4s", buf, p)
            hdr = 8
            if size == 1:
                if p + 16 > end:
                    return
                (size,) = struct.unpack_from(">Q", buf, p + 8)
                hdr = 16
            elif size == 0:
                size = end - p
            if size < hdr or p + size > end:
                raise ValueError(f"box {cc!r} at byte {p} runs past its parent ({size} bytes)")
            yield cc, p + hdr, p + size
            p += size

    def _find(self, start: int, end: int, *path: bytes):
        for cc, s, e in self._boxes(start, end):
            if cc == path[0]:
                return (s, e) if len(path) == 1 else self._find(s, e, *path[1:])
        return None

    def _track(self, s: int, e: int) -> bool:
        buf = self.buf
        mdia = self._find(s, e, b"mdia")
        hdlr = mdia and self._find(*mdia, b"hdlr")
        if not hdlr or bytes(buf[hdlr[0] + 8:hdlr[0] + 12]) != b"vide":
            return False
        mdhd = self._find(*mdia, b"mdhd")
        if mdhd:
            if buf[mdhd[0]] == 1:
                self.timescale, self.duration = struct.unpack_from(">IQ", buf, mdhd[0] + 20)
            else:
                self.timescale, self.duration = struct.unpack_from(">II", buf, mdhd[0] + 12)
        stbl = self._find(*mdia, b"minf", b"stbl")
        if stbl is None:
            raise ValueError("video track without a sample table")
        tab = {cc: (bs, be) for cc, bs, be in self._boxes(*stbl)}
        # ---- stsd: the first sample entry ----
        bs, be = tab[b"stsd"]
        (n_entries,) = struct.unpack_from(">I", buf, bs + 4)
        if n_entries < 1:
            raise ValueError("empty sample description")
        esize, self.codec = struct.unpack_from(">I4s", buf, bs + 8)
        ent = bs + 8
        self.width, self.height = struct.unpack_from(">HH", buf, ent + 32)
        (self.depth,) = struct.unpack_from(">H", buf, ent + 82)
        if self.codec == b"mp4v":
            esds = self._find(ent + 86, ent + esize, b"esds")
            if esds:
                self.oti = self._object_type(esds[0] + 4, esds[1])
        # ---- stsz / stz2: sample sizes ----
        if b"stsz" in tab:
            bs, _ = tab[b"stsz"]
            uniform, n = struct.unpack_from(">II", buf, bs + 4)
            sizes = np.full(n, uniform, np.int64) if uniform else np.frombuffer(buf, ">u4", n, bs + 12).astype(np.int64)
        elif b"stz2" in tab:
            bs, _ = tab[b"stz2"]
            field, n = buf[bs + 7], struct.unpack_from(">I", buf, bs + 8)[0]
            if field == 4:
                raw = np.frombuffer(buf, np.uint8, (n + 1) // 2, bs + 12)
                sizes = np.stack([raw >> 4, raw & 15], 1).reshape(-1)[:n].astype(np.int64)
            else:
                sizes = np.frombuffer(buf, {8: ">u1", 16: ">u2"}[field], n, bs + 12).astype(np.int64)
        else:
            raise ValueError("sample table without stsz / stz2")
        n = len(sizes)
        # ---- chunk offsets, samples per chunk ----
        if b"stco" in tab:
            bs, _ = tab[b"stco"]
            offs = np.frombuffer(buf, ">u4", struct.unpack_from(">I", buf, bs + 4)[0], bs + 8).astype(np.int64)
        elif b"co64" in tab:
            bs, _ = tab[b"co64"]
            offs = np.frombuffer(buf, ">u8", struct.unpack_from(">I", buf, bs + 4)[0], bs + 8).astype(np.int64)
        else:
            raise ValueError("sample table without stco / co64")
        bs, _ = tab[b"stsc"]
        runs = np.frombuffer(buf, ">u4", 3 * struct.unpack_from(">I", buf, bs + 4)[0], bs + 8).reshape(-1, 3).astype(np.int64)
        per_chunk = np.zeros(len(offs), np.int64)
        for i, (first, spc, _desc) in enumerate(runs):
            last = runs[i + 1][0] - 1 if i + 1 < len(runs) else len(offs)
            per_chunk[first - 1:last] = spc
        if per_chunk.sum() < n:
            raise ValueError(f"stsc / stco describe {int(per_chunk.sum())} samples, stsz {n}")
        # sample k of chunk c starts at offs[c] + the sizes of the chunk's earlier samples
        chunk_of = np.repeat(np.arange(len(offs)), per_chunk)[:n]
        first_of = np.concatenate([[0], np.cumsum(per_chunk)])[:-1]
        csum = np.concatenate([[0], np.cumsum(sizes)])
        starts = offs[chunk_of] + csum[:n] - csum[first_of[chunk_of]]
        if n and int((starts + sizes).max()) > len(buf):
            raise ValueError("sample table points past the end of the file (truncated download?)")
        self.frames = list(zip(starts.tolist(), sizes.tolist()))
        # ---- stts: frame rate = samples per second of media time ----
        if b"stts" in tab and self.timescale:
            bs, _ = tab[b"stts"]
            ent_n = struct.unpack_from(">I", buf, bs + 4)[0]
            tt = np.frombuffer(buf, ">u4", 2 * ent_n, bs + 8).reshape(-1, 2).astype(np.int64)
            ticks = int((tt[:, 0] * tt[:, 1]).sum())
            if ticks:
                self._fps = int(tt[:, 0].sum()) * self.timescale / ticks
        if not self._fps and self.duration and self.timescale:
            self._fps = n * self.timescale / self.duration
        return True

    def _object_type(self, p: int, end: int):
        """objectTypeIndication of an esds: ES_Descriptor (tag 3) > DecoderConfigDescriptor (tag 4); descriptor lengths are 1-4 bytes, 7 bits each"""
        buf = self.buf

        def tag_len(q):
            tag, q, ln = buf[q], q + 1, 0
            for _ in range(4):
                b = buf[q]
                q += 1
                ln = (ln << 7) | (b & 0x7F)
                if not b & 0x80:
                    break
            return tag, ln, q
        tag, _, q = tag_len(p)
        if tag != 3:
            return None
        flags = buf[q + 2]
        q += 3 + (2 if flags & 0x80 else 0)                 # ES_ID, flags, dependsOn_ES_ID
        if flags & 0x40:
            q += 1 + buf[q]                                 # URL string
        if flags & 0x20:
            q += 2                                          # OCR_ES_Id
        if q >= end:
            return None
        tag, _, q = tag_len(q)
        return int(buf[q]) if tag == 4 and q < end else None

    @property
    def fps(self) -> float:
        return self._fps or 25.0


def _yuv_to_rgb(y: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    """BT.601 limited-range Y'CbCr (what YUV4MPEG2 carries by convention) -> uint8 RGB, round to nearest"""
    yf = (y.astype(np.float32) - 16.0) * (255.0 / 219.0)
    uf, vf = u.astype(np.float32) - 128.0, v.astype(np.float32) - 128.0
    k = 255.0 / 224.0
    r = yf + 1.402 * k * vf
    g = yf - (0.344136 * k) * uf - (0.714136 * k) * vf
    b = yf + 1.772 * k * uf
    return np.clip(np.rint(np.stack([r, g, b], -1)), 0, 255).astype(np.uint8)


class VideoReader:
    """decord.VideoReader's call surface over .avi (MJPG / uncompressed BGR24) and .y4m (C420*, C422, C444, 8 bit) files"""

    def __init__(self, path: str, ctx=None, num_threads: int = 1):
        self.path = path
        self._threads = max(1, int(num_threads or 1))        # decord: 0 = automatic; here 0 / None = one thread
        with open(path, "rb") as f:
            try:                                            # map, do not read: an hour of Motion-JPEG is gigabytes, and only the frames asked for are touched
                import mmap
                self._data = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            except (ValueError, OSError):                   # an empty file, a file system without mmap
                self._data = f.read()
        head = bytes(self._data[:12])
        if head.startswith(b"YUV4MPEG2"):
            self._init_y4m()
        elif head[:4] == b"RIFF" and head[8:12] != b"WEBP":
            self._init_avi()
        elif head[4:8] in (b"ftyp", b"moov", b"mdat", b"wide", b"free", b"skip", b"pnot"):
            self._init_bmff()
        elif head[:6] in (b"GIF87a", b"GIF89a") or head[:8] == b"\x89PNG\r\n\x1a\n" or head[:4] in (b"II*\0", b"MM\0*") or (head[:4] == b"RIFF" and head[8:12] == b"WEBP"):
            self._init_pil()
        else:
            raise ValueError(f"{path}: not a RIFF AVI, YUV4MPEG2, ISO base media / QuickTime or animated-image file (what the built-in reader covers; install "
                             f"decord / PyAV for anything else)")

    # ---- AVI ------------------------------------------------------------------------------------------------------------------
    def _init_avi(self):
        ix = _AviIndex(self._data)
        comp = ix.compression.upper()
        if comp in (b"MJPG", b"JPEG") or ix.handler.upper() in (b"MJPG",) and comp not in (b"\0\0\0\0", b"DIB "):
            self._kind = "mjpg"
        elif comp in (b"\0\0\0\0", b"DIB ", b"RGB ") and ix.bit_count == 24:
            self._kind = "bgr24"
        else:
            raise ValueError(f"{self.path}: AVI video stream {ix.compression!r}/{ix.handler!r} ({ix.bit_count} bit) needs a codec; the built-in reader "
                             f"decodes MJPG and uncompressed 24-bit frames only")
        self._ix, self._fps = ix, ix.fps
        self._n, self.width, self.height = len(ix.frames), ix.width, ix.height

    def _avi_frame(self, i: int) -> np.ndarray:
        off, size = self._ix.frames[i]
        if self._kind == "mjpg":
            from PIL import Image
            if size == 0:                                   # a dropped frame repeats its predecessor (AVI convention)
                return self._avi_frame(i - 1) if i else np.zeros((self.height, self.width, 3), np.uint8)
            return np.asarray(Image.open(io.BytesIO(self._data[off:off + size])).convert("RGB"))
        stride = (self.width * 3 + 3) & ~3                  # DIB rows are padded to 4 bytes, stored bottom-up unless the height is negative
        rows = np.frombuffer(self._data, np.uint8, stride * self.height, off).reshape(self.height, stride)[:, :self.width * 3]
        img = rows.reshape(self.height, self.width, 3)[:, :, ::-1]
        return np.ascontiguousarray(img if self._ix.top_down else img[::-1])

    # ---- ISO base media / QuickTime with individually coded frames (round 6) ----------------------------------------------------------------
    def _init_bmff(self):
        ix = _BmffIndex(self._data)
        cc = ix.codec
        if cc in (b"jpeg", b"mjpa", b"MJPG", b"mjpg", b"dmb1", b"AVDJ") or (cc == b"mp4v" and ix.oti == 0x6C):
            self._kind = "bmff_pil"                          # every sample a JPEG (Photo-JPEG / Motion-JPEG A: the first field is a complete JPEG)
        elif cc in (b"png ", b"MPNG") or (cc == b"mp4v" and ix.oti == 0x6D):
            self._kind = "bmff_pil"
        elif cc == b"raw " and ix.depth == 24:
            self._kind = "bmff_rgb24"                        # QuickTime uncompressed: packed R G B, top-down, rows not padded
            bad = [i for i, (_, size) in enumerate(ix.frames) if size != ix.width * ix.height * 3]
            if bad:
                raise ValueError(f"{self.path}: 'raw ' sample {bad[0]} is not {ix.width} x {ix.height} x 3 bytes")
        else:
            raise CodecNeeded(f"{self.path}: the video track is {cc.decode('latin-1')!r} ({CODEC_NAMES.get(cc, 'a coded format')}"
                              f"{'' if ix.oti is None else f', objectTypeIndication 0x{ix.oti:02X}'}), {len(ix.frames)} frames {ix.width}x{ix.height} at "
                              f"{ix.fps:.3f} fps: it needs a decoder (decord or PyAV); the built-in reader decodes Motion-JPEG / PNG / uncompressed tracks only")
        self._ix, self._fps = ix, ix.fps
        self._n, self.width, self.height = len(ix.frames), ix.width, ix.height

    def _bmff_frame(self, i: int) -> np.ndarray:
        off, size = self._ix.frames[i]
        if self._kind == "bmff_rgb24":
            return np.frombuffer(self._data, np.uint8, size, off).reshape(self.height, self.width, 3).copy()
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(self._data[off:off + size])).convert("RGB"))

    # ---- animated images Pillow decodes itself: GIF, APNG, animated WebP, multi-page TIFF (round 6) ----------------------------------------
    def _init_pil(self):
        """frame-sequence containers whose codecs ship INSIDE Pillow (LZW / deflate / VP8 via libwebp): the last kind of "video" file the image can
        read without a decoder library.  Frame rate from the per-frame `duration` (ms) the container records; one still image is a 1-frame video."""
        from PIL import Image
        im = Image.open(io.BytesIO(self._data))
        self._pil = im
        self._n = int(getattr(im, "n_frames", 1))
        self.width, self.height = im.size
        durs = []
        for i in range(min(self._n, 64)):                   # (the average over the first frames: GIFs may vary it per frame)
            im.seek(i)
            d = im.info.get("duration")
            if d:
                durs.append(float(d))
        self._fps = 1000.0 / (sum(durs) / len(durs)) if durs else 25.0
        self._kind, self._pil_at = "pil", -1

    def _pil_frame(self, i: int) -> np.ndarray:
        im = self._pil
        if i < self._pil_at:                                # delta-coded formats compose forward: Pillow replays from the start on a backward seek
            im.seek(0)
        im.seek(i)
        self._pil_at = i
        return np.asarray(im.convert("RGB"))

    # ---- YUV4MPEG2 ------------------------------------------------------------------------------------------------------------
    def _init_y4m(self):
        end = self._data.find(b"\n")
        if end < 0:
            raise ValueError(f"{self.path}: YUV4MPEG2 header without a line end")
        toks = self._data[:end].decode("ascii").split(" ")[1:]
        self._chroma, num, den = "420jpeg", 25, 1
        for t in toks:
            if t[0] == "W":
                self.width = int(t[1:])
            elif t[0] == "H":
                self.height = int(t[1:])
            elif t[0] == "F":
                num, den = (int(x) for x in t[1:].split(":"))
            elif t[0] == "C":
                self._chroma = t[1:]
        if self._chroma.startswith("420"):
            self._cw, self._ch = (self.width + 1) // 2, (self.height + 1) // 2
        elif self._chroma == "422":
            self._cw, self._ch = (self.width + 1) // 2, self.height
        elif self._chroma == "444":
            self._cw, self._ch = self.width, self.height
        else:
            raise ValueError(f"{self.path}: YUV4MPEG2 colour space C{self._chroma} is not covered (8-bit C420* / C422 / C444 are)")
        self._fps = num / den
        self._fsize = self.width * self.height + 2 * self._cw * self._ch
        self._offsets, p = [], end + 1
        while p < len(self._data):
            if self._data[p:p + 5] != b"FRAME":
                raise ValueError(f"{self.path}: YUV4MPEG2 frame marker expected at byte {p}")
            nl = self._data.find(b"\n", p)
            if nl < 0:
                break                                       # truncated inside a frame header
            p = nl + 1
            if p + self._fsize > len(self._data):
                break                                       # truncated last frame
            self._offsets.append(p)
            p += self._fsize
        self._kind, self._n = "y4m", len(self._offsets)

    def _y4m_frame(self, i: int) -> np.ndarray:
        p, w, h, cw, ch = self._offsets[i], self.width, self.height, self._cw, self._ch
        y = np.frombuffer(self._data, np.uint8, w * h, p).reshape(h, w)
        u = np.frombuffer(self._data, np.uint8, cw * ch, p + w * h).reshape(ch, cw)
        v = np.frombuffer(self._data, np.uint8, cw * ch, p + w * h + cw * ch).reshape(ch, cw)
        if (ch, cw) != (h, w):                              # nearest-neighbour chroma upsampling (what a decoder's fast path does)
            u = np.repeat(np.repeat(u, -(-h // ch), 0), -(-w // cw), 1)[:h, :w]
            v = np.repeat(np.repeat(v, -(-h // ch), 0), -(-w // cw), 1)[:h, :w]
        return _yuv_to_rgb(y, u, v)

    # ---- decord's surface -----------------------------------------------------------------------------------------------------
    def __len__(self) -> int:
        return self._n

    def get_avg_fps(self) -> float:
        return float(self._fps)

    def _frame(self, i: int) -> np.ndarray:
        i = int(i)
        if not -self._n <= i < self._n:
            raise IndexError(f"frame {i} of {self._n}")
        i %= self._n
        if self._kind == "pil":
            return self._pil_frame(i)
        if self._kind.startswith("bmff"):
            return self._bmff_frame(i)
        return self._y4m_frame(i) if self._kind == "y4m" else self._avi_frame(i)

    def __getitem__(self, i: int) -> _Batch:
        return _Batch(self._frame(i))

    def get_batch(self, indices: Sequence[int]) -> _Batch:
        """decord's get_batch.  Individually coded frames (Motion-JPEG / PNG samples) decode on `num_threads` threads, as decord's reader does with its own
        (Pillow's decoders release the GIL); frame order is the order of `indices`.  Delta-coded animated images compose forward and stay sequential."""
        indices = [int(i) for i in indices]
        if not indices:
            return _Batch(np.zeros((0, self.height, self.width, 3), np.uint8))
        parallel = self._threads > 1 and len(indices) > 1 and (self._kind == "bmff_pil" or (self._kind == "mjpg"))
        if parallel:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(self._threads, len(indices))) as pool:
                frames = list(pool.map(self._frame, indices))
        else:
            frames = [self._frame(i) for i in indices]
        return _Batch(np.stack(frames))

    def seek(self, pos: int) -> None:                       # random access needs no state
        return None


def cpu(_index: int = 0):
    """decord.cpu(0) stand-in (the context argument is ignored: this reader is host code)"""
    return None


# ---- writers: what the tests (and anyone without a video tool) use to make a real file -------------------------------------------------
def write_avi(path: str, frames: np.ndarray, fps: float = 30.0, codec: str = "MJPG", quality: int = 92) -> None:
    """uint8 [T, H, W, 3] RGB -> AVI 1.0 with an idx1 index; codec "MJPG" (Pillow's baseline JPEG per frame) or "DIB " (uncompressed BGR24)"""
    frames = np.asarray(frames, np.uint8)
    T, H, W, _ = frames.shape
    payloads = []
    for fr in frames:
        if codec == "MJPG":
            from PIL import Image
            b = io.BytesIO()
            Image.fromarray(fr).save(b, "JPEG", quality=quality, subsampling=0)
            payloads.append(b.getvalue())
        else:
            stride = (W * 3 + 3) & ~3
            rows = np.zeros((H, stride), np.uint8)
            rows[:, :W * 3] = fr[::-1, :, ::-1].reshape(H, W * 3)
            payloads.append(rows.tobytes())
    scale = 1000
    rate = int(round(fps * scale))

    def chunk(cc: bytes, body: bytes) -> bytes:
        return cc + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    def lst(kind: bytes, body: bytes) -> bytes:
        return b"LIST" + struct.pack("<I", len(body) + 4) + kind + body
    biggest = max((len(p) for p in payloads), default=0)
    avih = struct.pack("<14I", int(round(1e6 / fps)), 0, 0, 0x10, T, 0, 1, biggest, W, H, 0, 0, 0, 0)
    comp = b"MJPG" if codec == "MJPG" else b"\0\0\0\0"
    strh = b"vids" + (b"MJPG" if codec == "MJPG" else b"DIB ") + struct.pack("<IHHIIIIIIII4H", 0, 0, 0, 0, scale, rate, 0, T, biggest, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHH", 40, W, H, 1, 24) + comp + struct.pack("<IiiII", biggest if codec == "MJPG" else ((W * 3 + 3) & ~3) * H, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    tag = b"00dc" if codec == "MJPG" else b"00db"
    movi_body, idx, pos = b"", b"", 4
    for p in payloads:
        idx += tag + struct.pack("<III", 0x10, pos, len(p))
        c = chunk(tag, p)
        movi_body += c
        pos += len(c)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def write_mov(path: str, frames: np.ndarray, fps: float = 30.0, codec: str = "jpeg", quality: int = 92, per_chunk: int = 3, co64: bool = False) -> None:
    """uint8 [T, H, W, 3] RGB -> a QuickTime / ISO base media file with ONE video track of individually coded samples: codec "jpeg" (QuickTime
    Photo-JPEG), "png ", "raw " (packed RGB24), or "mp4v-jpeg" (how MP4 muxers label Motion-JPEG: an 'mp4v' entry whose esds says
    objectTypeIndication 0x6C).  `per_chunk` samples per chunk (the last chunk may be short: two stsc runs), 32- or 64-bit chunk offsets."""
    from PIL import Image
    frames = np.asarray(frames, np.uint8)
    T, H, W, _ = frames.shape
    payloads = []
    for fr in frames:
        if codec == "raw ":
            payloads.append(fr.tobytes())
        else:
            b = io.BytesIO()
            if codec == "png ":
                Image.fromarray(fr).save(b, "PNG")
            else:
                Image.fromarray(fr).save(b, "JPEG", quality=quality, subsampling=0)
            payloads.append(b.getvalue())

    def box(cc: bytes, body: bytes) -> bytes:
        return struct.pack(">I4s", 8 + len(body), cc) + body

    def full(cc: bytes, body: bytes, version: int = 0, flags: int = 0) -> bytes:
        return box(cc, struct.pack(">I", (version << 24) | flags) + body)
    timescale = 30000
    delta = int(round(timescale / fps))
    ftyp = box(b"ftyp", (b"isom" if codec == "mp4v-jpeg" else b"qt  ") + struct.pack(">I", 0x200) + (b"isommp41" if codec == "mp4v-jpeg" else b"qt  "))
    mdat_off = len(ftyp) + 8
    offs, pos = [], mdat_off
    for c in range(0, T, per_chunk):
        offs.append(pos)
        pos += sum(len(p) for p in payloads[c:c + per_chunk])
    mdat = box(b"mdat", b"".join(payloads))
    fourcc = b"mp4v" if codec == "mp4v-jpeg" else codec.encode("latin-1")
    entry_body = (b"\0" * 6 + struct.pack(">H", 1) + struct.pack(">HHIII", 0, 0, 0, 0, 512) + struct.pack(">HH", W, H) + struct.pack(">II", 0x480000, 0x480000)
                  + struct.pack(">IH", 0, 1) + b"\0" * 32 + struct.pack(">Hh", 24, -1))
    if codec == "mp4v-jpeg":
        dcfg = bytes([0x6C, 0x11]) + b"\0\0\0" + struct.pack(">II", 0, 0)                     # objectTypeIndication 0x6C = JPEG, visual stream
        es = struct.pack(">HB", 1, 0) + bytes([4, len(dcfg)]) + dcfg + bytes([6, 1, 2])
        entry_body += full(b"esds", bytes([3, len(es)]) + es)
    stsd = full(b"stsd", struct.pack(">I", 1) + box(fourcc, entry_body))
    stts = full(b"stts", struct.pack(">III", 1, T, delta))
    n_chunks = len(offs)
    runs = [(1, min(per_chunk, T), 1)] if T else []
    if T % per_chunk and n_chunks > 1:
        runs.append((n_chunks, T % per_chunk, 1))
    stsc = full(b"stsc", struct.pack(">I", len(runs)) + b"".join(struct.pack(">III", *r) for r in runs))
    stsz = full(b"stsz", struct.pack(">II", 0, T) + b"".join(struct.pack(">I", len(p)) for p in payloads))
    stco = full(b"co64" if co64 else b"stco", struct.pack(">I", n_chunks) + b"".join(struct.pack(">Q" if co64 else ">I", o) for o in offs))
    stbl = box(b"stbl", stsd + stts + stsc + stsz + stco)
    dinf = box(b"dinf", full(b"dref", struct.pack(">I", 1) + full(b"url ", b"", flags=1)))
    minf = box(b"minf", full(b"vmhd", struct.pack(">HHHH", 0, 0, 0, 0), flags=1) + dinf + stbl)
    hdlr = full(b"hdlr", struct.pack(">I4s", 0, b"vide") + b"\0" * 12 + b"VideoHandler\0")
    mdhd = full(b"mdhd", struct.pack(">IIIIHH", 0, 0, timescale, T * delta, 0x55C4, 0))
    mdia = box(b"mdia", mdhd + hdlr + minf)
    matrix = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    tkhd = full(b"tkhd", struct.pack(">IIIII", 0, 0, 1, 0, T * delta) + b"\0" * 8 + struct.pack(">HHHH", 0, 0, 0, 0) + matrix + struct.pack(">II", W << 16, H << 16), flags=3)
    mvhd = full(b"mvhd", struct.pack(">IIIIIH", 0, 0, timescale, T * delta, 0x10000, 0x100) + b"\0" * 10 + matrix + b"\0" * 24 + struct.pack(">I", 2))
    moov = box(b"moov", mvhd + box(b"trak", tkhd + mdia))
    with open(path, "wb") as f:
        f.write(ftyp + mdat + moov)


def rgb_to_yuv420(frame: np.ndarray):
    """uint8 RGB [H, W, 3] -> BT.601 limited-range planes (Y [H, W], U and V [ceil(H/2), ceil(W/2)], 2x2 box-averaged chroma)"""
    f = frame.astype(np.float32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    u = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    v = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    H, W = y.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2

    def sub(p):
        p = np.pad(p, ((0, Hp - H), (0, Wp - W)), mode="edge")
        return p.reshape(Hp // 2, 2, Wp // 2, 2).mean((1, 3))
    q = lambda p: np.clip(np.rint(p), 0, 255).astype(np.uint8)      # noqa: E731
    return q(y), q(sub(u)), q(sub(v))


def write_y4m(path: str, frames: np.ndarray, fps: Tuple[int, int] = (30, 1)) -> None:
    """uint8 [T, H, W, 3] RGB -> YUV4MPEG2, C420jpeg"""
    frames = np.asarray(frames, np.uint8)
    T, H, W, _ = frames.shape
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F{fps[0]}:{fps[1]} Ip A1:1 C420jpeg\n".encode("ascii"))
        for fr in frames:
            y, u, v = rgb_to_yuv420(fr)
            f.write(b"FRAME\n" + y.tobytes() + u.tobytes() + v.tobytes())
