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
