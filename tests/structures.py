"""Seeded random batch structures shared by the generator that runs THE REFERENCE on them (tests/golden/make_random_structures.py -> random_structures.npz),
the CPU test that pins the oracle to those outputs and the GPU test that holds the HIP path to them.  Pure Python + torch's generator: the same draw everywhere."""
import torch


def draw(seed: int) -> dict:
    """batch of 1-3 prompts; padding on either side (the mask decides, eagle_archv1.py:272); 0-6 text tokens before each <image>, 0-9 after; clips of
    1-11 frames; optionally two clips in the first prompt - the head of prompt b reshapes by images[b].shape[0], `images` indexed by the BATCH ROW
    (grounding_qwen2.py:153), so then every clip of the batch gets one length for the reference itself to run; a token budget that pools to 1 x 1 ... 3 x 3
    or not at all; image side 56 or 70 (4 x 4 / 5 x 5 patches)"""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())       # noqa: E731
    side = (56, 70)[ri(0, 1)]
    budget = (16, 40, 72, 200, 100000)[ri(0, 4)]
    vocab = 320                                                             # tiny_config's
    B = ri(1, 3)
    two_first = ri(0, 2) == 0
    T, rows = [], []
    t_all = ri(1, 11)
    for b in range(B):
        t = t_all if two_first else ri(1, 11)
        n_img = 2 if (b == 0 and two_first) else 1
        T += [t] * n_img
        toks = []
        for _ in range(n_img):
            toks += [ri(1, vocab - 1) for _ in range(ri(0, 6))] + [-200]
        toks += [ri(1, vocab - 1) for _ in range(ri(0, 9))]
        rows.append(toks)
    L = max(len(r) for r in rows) + ri(0, 2)
    ids = torch.zeros((B, L), dtype=torch.long)
    mask = torch.zeros((B, L), dtype=torch.bool)
    for b, r in enumerate(rows):
        off = (L - len(r)) if ri(0, 1) else 0                               # left- or right-padded
        ids[b, off:off + len(r)] = torch.tensor(r)
        mask[b, off:off + len(r)] = True
    return dict(seed=seed, side=side, budget=budget, B=B, T=T, two_first=two_first, ids=ids, mask=mask, L=L,
                what=f"seed {seed}: B={B} T={T} L={L} side={side} budget={budget} two_first={two_first}")


def config(s: dict):
    from videoitg_amd.config import tiny_config
    return tiny_config(vision_token_num=s["budget"], vision={"image_size": s["side"]})


def images(s: dict):
    from videoitg_amd import synth
    return [synth.synth_pixel_values(t, s["side"], seed=77 + 13 * s["seed"] + i) for i, t in enumerate(s["T"])]


N_STRUCTURES = 16
