"""Row A2: read_video_decord / read_video_pyav / the plugin's load_video EXECUTE against fake decoder modules
(decord and PyAV are not in this image - the reference imports them at eagle/mm_utils.py:27-29).  The fakes implement the
few calls the readers make (VideoReader(len, get_avg_fps, get_batch, seek); av.open / streams / decode / demux) over a
synthetic 'video' whose frame i is filled with the value i, so the test can check WHICH frames were decoded."""
import sys
import types

import numpy as np
import pytest

from videoitg_amd import mm_utils

H, W = 6, 8


def frame(i):
    return np.full((H, W, 3), i % 251, dtype=np.uint8)


class FakeBatch:
    def __init__(self, arr):
        self.arr = arr

    def asnumpy(self):
        return self.arr


class FakeVideoReader:
    total, fps = 900, 30.0

    def __init__(self, path, ctx=None, num_threads=1):
        self.path, self.seeked = path, None

    def __len__(self):
        return self.total

    def get_avg_fps(self):
        return self.fps

    def get_batch(self, idx):
        return FakeBatch(np.stack([frame(i) for i in idx]))

    def seek(self, pos):
        self.seeked = pos


@pytest.fixture
def fake_decord(monkeypatch):
    m = types.ModuleType("decord")
    m.VideoReader, m.cpu = FakeVideoReader, lambda *a, **k: None
    monkeypatch.setitem(sys.modules, "decord", m)
    return m


class FakeFrame:
    def __init__(self, i):
        self.i = i

    def to_ndarray(self, format="rgb24"):
        assert format == "rgb24"
        return frame(self.i)


class FakePacket:
    def __init__(self, frames):
        self.frames = frames

    def decode(self):
        return self.frames


class FakeContainer:
    def __init__(self, path, total, fps, report_frames=True):
        cc = types.SimpleNamespace(thread_count=1)
        st = types.SimpleNamespace(thread_type=None, codec_context=cc, average_rate=fps, frames=total if report_frames else 0)
        self.streams = types.SimpleNamespace(video=[st])
        self.total, self.decoded, self.closed = total, 0, False

    def decode(self, video=0):
        for i in range(self.total):
            self.decoded += 1
            yield FakeFrame(i)

    def demux(self, video=0):
        for s in range(0, self.total, 7):
            yield FakePacket([FakeFrame(i) for i in range(s, min(s + 7, self.total))])

    def seek(self, pos):
        pass

    def close(self):
        self.closed = True


@pytest.fixture
def fake_av(monkeypatch):
    m = types.ModuleType("av")
    m.opened = []

    def _open(path):
        c = FakeContainer(path, 900, 30.0, report_frames=not path.endswith(".webm"))
        m.opened.append(c)
        return c
    m.open = _open
    monkeypatch.setitem(sys.modules, "av", m)
    return m


def test_read_video_decord_samples_the_reference_indices(fake_decord):
    frames, idx, secs = mm_utils.read_video_decord("x.mp4", num_frm=16, target_fps=2)        # infer.py:44-51
    want = mm_utils.get_frame_indices(900, 30.0, 2, 16)
    assert idx == want and secs == 30 and frames.shape == (16, H, W, 3) and frames.dtype == np.uint8
    assert [int(f[0, 0, 0]) for f in frames] == [i % 251 for i in want]
    few, idx2, _ = mm_utils.read_video_decord("x.mp4", num_frm=512, target_fps=2)           # fewer candidates than requested
    assert len(idx2) == len(range(0, 900, 15)) == few.shape[0]


def test_read_video_pyav_fast_and_fallback_paths(fake_av):
    a = mm_utils.read_video_pyav("x.mp4", num_frm=16, target_fps=2)                           # eagle/mm_utils.py:43-69
    want = mm_utils.get_frame_indices(900, 30.0, 2, 16)
    assert a.shape == (16, H, W, 3) and [int(f[0, 0, 0]) for f in a] == [i % 251 for i in want]
    assert fake_av.opened[-1].decoded == want[-1] + 2          # sequential decode stops right after the last wanted frame
    b = mm_utils.read_video_pyav("x.webm", num_frm=16, target_fps=2)                          # webm: demux everything, then sample
    assert np.array_equal(a, b)


def test_plugin_load_video_uses_decord_then_pyav(fake_decord, fake_av, monkeypatch):
    from videoitg_amd.lmms_plugin.models import videoitg as plug
    imgs, idx = plug.load_video("x.mp4", num_frames=8, target_fps=1)                        # videoitg.py:95-130
    want = mm_utils.get_seq_frames(900, 30.0, 1, 8)
    assert list(idx) == want and len(imgs) == 8 and imgs[0].size == (W, H)
    assert [im.getpixel((0, 0))[0] for im in imgs] == [i % 251 for i in want]
    monkeypatch.setitem(sys.modules, "decord", None)           # decord missing -> ImportError -> PyAV fallback
    imgs2, idx2 = plug.load_video("x.mp4", num_frames=8, target_fps=1)
    assert list(idx2) == want and [im.getpixel((0, 0))[0] for im in imgs2] == [i % 251 for i in want]
    assert fake_av.opened[-1].closed


# ---- real files: the built-in reader (videoitg_amd/video_io.py) behind the same entry points, decord absent ------------------------------
def _clip(T=75, h=36, w=48):
    """frames whose content says which frame it is: a flat grey level per frame (survives JPEG) + a moving bright block"""
    fr = np.zeros((T, h, w, 3), np.uint8)
    for i in range(T):
        fr[i] = 20 + 3 * (i % 70)
        fr[i, 4:12, (i % (w - 8)):(i % (w - 8)) + 8] = (250, 40, 40)
    return fr


@pytest.fixture
def no_decoders(monkeypatch):
    monkeypatch.setitem(sys.modules, "decord", None)
    monkeypatch.setitem(sys.modules, "av", None)


def test_avi_uncompressed_round_trip_is_exact(tmp_path, no_decoders):
    from videoitg_amd import video_io
    clip = _clip(T=33, h=35, w=47)                              # odd sizes: DIB rows are padded to 4 bytes
    path = str(tmp_path / "clip_raw.avi")
    video_io.write_avi(path, clip, fps=12.5, codec="DIB ")
    vr = video_io.VideoReader(path)
    assert len(vr) == 33 and abs(vr.get_avg_fps() - 12.5) < 1e-6 and (vr.height, vr.width) == (35, 47)
    assert np.array_equal(vr.get_batch(list(range(33))).asnumpy(), clip)
    assert np.array_equal(vr[-1].asnumpy(), clip[-1])
    with pytest.raises(IndexError):
        vr[33]


def test_read_video_decord_on_a_real_mjpeg_avi(tmp_path, no_decoders):
    """infer.py:44-51 on a REAL file: 300 frames at 29.97 fps, MJPEG in AVI, decord absent -> the built-in reader; the sampled indices are
    get_frame_indices' and every returned frame is the frame of that index (grey level), within JPEG error"""
    from videoitg_amd import video_io
    clip = _clip(T=300)
    path = str(tmp_path / "clip.avi")
    video_io.write_avi(path, clip, fps=29.97, codec="MJPG")
    frames, idx, secs = mm_utils.read_video_decord(path, num_frm=16, target_fps=2)
    want = mm_utils.get_frame_indices(300, 29.97, 2, 16)
    assert idx == want and secs == round(300 / 29.97) and frames.shape == (16, 36, 48, 3) and frames.dtype == np.uint8
    for f, i in zip(frames, idx):
        assert abs(int(f[30, 5, 0]) - int(clip[i, 30, 5, 0])) <= 3 and abs(int(f[30, 40, 2]) - int(clip[i, 30, 40, 2])) <= 3, i
        assert np.abs(f.astype(int) - clip[i].astype(int)).mean() < 4.0
    all_, idx_all, _ = mm_utils.read_video_decord(path, num_frm=512, target_fps=2)           # fewer candidates than asked for
    assert idx_all == list(range(0, 300, 15)) and len(all_) == 20


def test_y4m_round_trip_and_plugin_loader(tmp_path, no_decoders):
    from videoitg_amd import video_io
    from videoitg_amd.lmms_plugin.models import videoitg as plug
    clip = _clip(T=90, h=34, w=46)
    clip[:, :, :, :] = clip[:, :, :, :1]                       # grey: survives 4:2:0 chroma subsampling and the limited-range round trip to +-2
    path = str(tmp_path / "clip.y4m")
    video_io.write_y4m(path, clip, fps=(30000, 1001))
    vr = video_io.VideoReader(path)
    assert len(vr) == 90 and abs(vr.get_avg_fps() - 30000 / 1001) < 1e-9
    got = vr.get_batch([0, 44, 89]).asnumpy()
    assert np.abs(got.astype(int) - clip[[0, 44, 89]].astype(int)).max() <= 2
    imgs, idx = plug.load_video(path, num_frames=8, target_fps=1)                            # lmms_eval/models/videoitg.py:95-130
    assert list(idx) == mm_utils.get_seq_frames(90, 30000 / 1001, 1, 8) and imgs[0].size == (46, 34)
    assert [abs(im.getpixel((2, 30))[0] - int(clip[i, 30, 2, 0])) <= 2 for im, i in zip(imgs, idx)] == [True] * len(idx)


def test_files_that_need_a_codec_still_raise_like_the_reference(tmp_path, no_decoders):
    from videoitg_amd import video_io
    with pytest.raises(ImportError):
        mm_utils.read_video_decord(str(tmp_path / "movie.mkv"), num_frm=4)                   # no decord, not a container the built-in reader parses
    bad = tmp_path / "not_a_video.avi"
    bad.write_bytes(b"hello, this is not RIFF")
    with pytest.raises(ValueError):
        video_io.VideoReader(str(bad))
    clip = _clip(T=3)
    path = str(tmp_path / "h264.avi")
    video_io.write_avi(path, clip, codec="DIB ")
    raw = bytearray(open(path, "rb").read())
    k = raw.index(b"strf") + 8 + 16
    raw[k:k + 4] = b"H264"                                      # a stream that needs a real codec
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="needs a codec"):
        video_io.VideoReader(path)


def test_opendml_avix_segments_are_read_not_truncated(tmp_path, no_decoders):
    """AVIs over 1 GiB are OpenDML files - a 'RIFF AVI ' segment followed by 'RIFF AVIX' segments with their own movi lists.  The
    reader walks every segment (it used to stop after the first and return a truncated frame list without an error)."""
    import struct
    from videoitg_amd import video_io
    clip = _clip(T=21, h=20, w=24)
    first, rest = str(tmp_path / "a.avi"), str(tmp_path / "b.avi")
    video_io.write_avi(first, clip[:9], fps=10.0, codec="DIB ")
    video_io.write_avi(rest, clip[9:], fps=10.0, codec="DIB ")
    raw = open(rest, "rb").read()
    k = raw.index(b"movi") - 8                                   # the second file's LIST 'movi' becomes the body of a continuation segment
    (lsize,) = struct.unpack_from("<I", raw, k + 4)
    movi = raw[k:k + 8 + lsize]
    path = str(tmp_path / "opendml.avi")
    with open(path, "wb") as f:
        f.write(open(first, "rb").read() + b"RIFF" + struct.pack("<I", 4 + len(movi)) + b"AVIX" + movi)
    vr = video_io.VideoReader(path)
    assert len(vr) == 21
    assert np.array_equal(vr.get_batch(list(range(21))).asnumpy(), clip)


def test_avi_with_a_real_codec_falls_through_to_pyav(tmp_path, fake_av, monkeypatch):
    """without decord, *.avi goes to the built-in reader, which raises ValueError for a stream that needs a codec; the plugin's
    loader must then try PyAV like the reference's bare `except:` does (lmms_eval/models/videoitg.py:95-130) - and, with PyAV absent too, raise the
    first reader's error, not a misleading ImportError about `av`."""
    from videoitg_amd import video_io
    from videoitg_amd.lmms_plugin.models import videoitg as plug
    monkeypatch.setitem(sys.modules, "decord", None)
    path = str(tmp_path / "h264.avi")
    video_io.write_avi(path, _clip(T=3), codec="DIB ")
    raw = bytearray(open(path, "rb").read())
    k = raw.index(b"strf") + 8 + 16
    raw[k:k + 4] = b"H264"
    open(path, "wb").write(bytes(raw))
    imgs, idx = plug.load_video(path, num_frames=8, target_fps=1)          # PyAV (the fake: 900 frames at 30 fps) got its chance
    assert list(idx) == mm_utils.get_seq_frames(900, 30.0, 1, 8) and len(imgs) == 8 and fake_av.opened[-1].closed
    monkeypatch.setitem(sys.modules, "av", None)
    with pytest.raises(ValueError, match="needs a codec"):
        plug.load_video(path, num_frames=8, target_fps=1)


@pytest.mark.parametrize("fmt,suffix,exact", [("GIF", ".gif", False), ("PNG", ".png", True), ("WEBP", ".webp", True), ("TIFF", ".tif", True)])
def test_animated_images_are_videos_too(tmp_path, no_decoders, fmt, suffix, exact):
    """round 6, A2 / N3 "as far as the image allows": GIF / APNG / animated WebP / multi-page TIFF are frame sequences whose codecs ship inside Pillow, so
    `read_video_decord` and the plugin's loader read them without decord or PyAV (lossless formats exactly; GIF through its 256-colour palette)"""
    from PIL import Image
    from videoitg_amd import video_io
    from videoitg_amd.lmms_plugin.models import videoitg as plug
    clip = _clip(T=40, h=32, w=44)
    clip[:, :, :, :] = clip[:, :, :, :1]                       # grey levels: inside any palette
    frames = [Image.fromarray(f) for f in clip]
    path = str(tmp_path / f"clip{suffix}")
    kw = dict(save_all=True, append_images=frames[1:])
    if fmt in ("GIF", "PNG", "WEBP"):
        kw.update(duration=40, loop=0)                          # 25 fps
    if fmt == "WEBP":
        kw.update(lossless=True)
    try:
        frames[0].save(path, format=fmt, **kw)
    except (KeyError, OSError, ValueError) as e:                # a Pillow build without that writer
        pytest.skip(f"Pillow cannot write {fmt} here: {e}")
    vr = video_io.VideoReader(path)
    assert len(vr) == 40 and (vr.height, vr.width) == (32, 44)
    if fmt != "TIFF":
        assert abs(vr.get_avg_fps() - 25.0) < 1e-6
    got = vr.get_batch([0, 17, 39, 5]).asnumpy()               # (a backward seek in a delta-coded format included)
    want = clip[[0, 17, 39, 5]]
    assert got.shape == want.shape and (np.array_equal(got, want) if exact else np.abs(got.astype(int) - want.astype(int)).max() <= 4)
    imgs, idx = plug.load_video(path, num_frames=8, target_fps=1)
    assert len(imgs) == len(idx) >= 1 and imgs[0].size == (44, 32)


@pytest.mark.parametrize("codec,kw,exact", [("raw ", {}, True), ("png ", {"co64": True}, True), ("jpeg", {"per_chunk": 2}, False),
                                            ("mp4v-jpeg", {"per_chunk": 50}, False)])
def test_quicktime_and_mp4_tracks_of_individually_coded_frames(tmp_path, no_decoders, codec, kw, exact):
    """Round 6: *.mov / *.mp4 are CONTAINERS; when the track's samples are individually coded (QuickTime Photo-JPEG 'jpeg', 'png ', uncompressed
    'raw ', or Motion-JPEG as MP4 muxers label it: 'mp4v' + esds objectTypeIndication 0x6C) the sample table (stsc runs x 32- / 64-bit chunk offsets x
    stsz) gives every frame's bytes and Pillow decodes them - read_video_decord runs end to end on such a file without decord / PyAV."""
    from videoitg_amd import video_io
    clip = _clip(T=40, h=34, w=46)
    path = str(tmp_path / ("clip.mp4" if codec == "mp4v-jpeg" else "clip.mov"))
    video_io.write_mov(path, clip, fps=30000 / 1001, codec=codec, **kw)
    vr = video_io.VideoReader(path)
    assert len(vr) == 40 and (vr.width, vr.height) == (46, 34) and abs(vr.get_avg_fps() - 30000 / 1001) < 1e-6
    frames, idx, seconds = mm_utils.read_video_decord(path, num_frm=8, target_fps=2)           # infer.py:44-51
    assert list(idx) == mm_utils.get_frame_indices(40, 30000 / 1001, 2, 8) and frames.shape == (len(idx), 34, 46, 3) and frames.dtype == np.uint8
    err = np.abs(frames.astype(int) - clip[list(idx)].astype(int))
    assert err.max() == 0 if exact else err.mean() < 3.0
    assert np.array_equal(vr[-1].asnumpy(), vr[39].asnumpy())


def test_mp4_with_a_coded_track_names_the_codec_and_leaves_it_to_pyav(tmp_path, fake_av, monkeypatch):
    """an 'avc1' track (what nearly every real .mp4 holds): the container still parses - frame count, size, rate - and the error is the reference's
    ImportError (eagle/mm_utils.py:27-29 without decord) with the codec's name in it; the plugin's loader then goes to PyAV as the reference does"""
    from videoitg_amd import video_io
    from videoitg_amd.lmms_plugin.models import videoitg as plug
    monkeypatch.setitem(sys.modules, "decord", None)
    path = str(tmp_path / "movie.mp4")
    video_io.write_mov(path, _clip(T=5), fps=25.0, codec="raw ")
    raw = bytearray(open(path, "rb").read())
    k = raw.index(b"stsd") + 4 + 8 + 4
    assert raw[k:k + 4] == b"raw "
    raw[k:k + 4] = b"avc1"
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ImportError, match=r"'avc1' \(H\.264\), 5 frames .* at 25\.000 fps"):
        mm_utils.read_video_decord(path, num_frm=4)
    imgs, idx = plug.load_video(path, num_frames=8, target_fps=1)          # PyAV (the fake: 900 frames at 30 fps)
    assert list(idx) == mm_utils.get_seq_frames(900, 30.0, 1, 8) and len(imgs) == 8
    monkeypatch.setitem(sys.modules, "av", None)
    with pytest.raises(ImportError, match="H.264"):
        plug.load_video(path, num_frames=8, target_fps=1)


def test_mp4_container_edge_cases(tmp_path, no_decoders):
    """moov before mdat with a 64-bit 'largesize' mdat header (files over 4 GiB use both), a truncated download, a file without a video track"""
    import struct
    from videoitg_amd import video_io
    clip = _clip(T=6, h=10, w=12)
    path = str(tmp_path / "a.mov")
    video_io.write_mov(path, clip, fps=12.0, codec="raw ", per_chunk=4)
    raw = open(path, "rb").read()
    ft = struct.unpack_from(">I", raw, 0)[0]
    md = struct.unpack_from(">I", raw, ft)[0]
    ftyp, mdat, moov = raw[:ft], raw[ft:ft + md], bytearray(raw[ft + md:])
    # re-mux: ftyp, moov, then mdat with a largesize header -> every chunk offset moves by len(moov) + 8
    k = moov.index(b"stco") + 4 + 4
    (n,) = struct.unpack_from(">I", moov, k)
    for i in range(n):
        (o,) = struct.unpack_from(">I", moov, k + 4 + 4 * i)
        struct.pack_into(">I", moov, k + 4 + 4 * i, o + len(moov) + 8)
    big = struct.pack(">I4sQ", 1, b"mdat", len(mdat) + 8) + mdat[8:]
    p2 = str(tmp_path / "faststart.mov")
    open(p2, "wb").write(ftyp + bytes(moov) + big)
    vr = video_io.VideoReader(p2)
    assert len(vr) == 6 and vr.get_avg_fps() == 12.0 and np.array_equal(vr.get_batch(list(range(6))).asnumpy(), clip)
    p3 = str(tmp_path / "cut.mov")
    open(p3, "wb").write(ftyp + bytes(moov) + big[:len(big) // 2])
    with pytest.raises(ValueError, match="past the end"):
        video_io.VideoReader(p3)
    p4 = str(tmp_path / "audio_only.mov")
    open(p4, "wb").write(raw.replace(b"vide", b"soun"))
    with pytest.raises(ValueError, match="no video track"):
        video_io.VideoReader(p4)
    p6 = str(tmp_path / "no_stsc.mov")
    open(p6, "wb").write(raw.replace(b"stsc", b"skip"))                   # a sample table without its sample-to-chunk box
    with pytest.raises(ValueError, match="malformed sample table"):
        video_io.VideoReader(p6)
    p5 = str(tmp_path / "no_moov.mp4")
    open(p5, "wb").write(ftyp + mdat)
    with pytest.raises(ValueError, match="no 'moov' box"):
        video_io.VideoReader(p5)


def test_get_batch_decodes_on_threads_in_index_order(tmp_path, no_decoders):
    """decord's `num_threads`: individually coded frames decode on a thread pool (Pillow's decoders release the GIL); the batch is in the order of the indices
    asked for - repeated and descending ones included - and equal to the one-thread result"""
    from videoitg_amd import video_io
    clip = _clip(T=30, h=34, w=46)
    idx = [29, 0, 7, 7, 3, 28, 1]
    for name, write in (("a.avi", lambda p: video_io.write_avi(p, clip, fps=30.0)), ("a.mov", lambda p: video_io.write_mov(p, clip, fps=30.0, codec="jpeg"))):
        path = str(tmp_path / name)
        write(path)
        one = video_io.VideoReader(path, num_threads=1).get_batch(idx).asnumpy()
        many = video_io.VideoReader(path, num_threads=4).get_batch(idx).asnumpy()
        assert np.array_equal(one, many) and np.array_equal(many[2], many[3])
        assert [int(np.abs(many[k].astype(int) - clip[i].astype(int)).mean() < 3.0) for k, i in enumerate(idx)] == [1] * len(idx)
    assert mm_utils.open_video_reader(path, num_threads=4)._threads == 4


def test_lossless_containers_round_trip_on_random_geometries(tmp_path, no_decoders):
    """odd widths (DIB rows pad to 4 bytes, QuickTime 'raw ' rows do not), 1-pixel frames, single-frame clips, chunking that does not divide the frame
    count: uncompressed AVI, raw / PNG QuickTime tracks return every byte they were given"""
    from videoitg_amd import video_io
    rng = np.random.default_rng(11)
    for k in range(24):
        T, H, W = int(rng.integers(1, 7)), int(rng.integers(1, 41)), int(rng.integers(1, 41))
        clip = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
        kind = ("dib", "raw ", "png ")[k % 3]
        path = str(tmp_path / f"c{k}.{'avi' if kind == 'dib' else 'mov'}")
        if kind == "dib":
            video_io.write_avi(path, clip, fps=float(rng.integers(1, 61)), codec="DIB ")
        else:
            video_io.write_mov(path, clip, fps=float(rng.integers(1, 61)), codec=kind, per_chunk=int(rng.integers(1, 5)), co64=bool(k & 1))
        vr = video_io.VideoReader(path, num_threads=int(rng.integers(1, 4)))
        assert len(vr) == T and (vr.height, vr.width) == (H, W), (kind, T, H, W)
        assert np.array_equal(vr.get_batch(list(range(T))).asnumpy(), clip), (kind, T, H, W)
        assert np.array_equal(vr[T - 1].asnumpy(), clip[-1])
