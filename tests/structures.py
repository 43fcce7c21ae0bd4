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


# ---- random ARCHITECTURES (round 6): widths, head counts, depths, GQA groups, FFN widths nobody chose -------------------------------------------------
N_ARCHS = 12


def draw_arch(seed: int) -> dict:
    """one prompt, one clip - the structure is not the point here - on a drawn architecture: tower of 1-4 heads of 72 (the SigLIP head size), FFN width any
    multiple of 4, 2-4 layers (hidden_states[-2] = all but the last), image side 56 / 62 / 70 / 84; decoder of 1-2 KV heads x GQA group 1 / 2 / 3 / 7 of
    head size 128, FFN width any multiple of 8, 1-3 layers, rope_theta 1e4 or 1e6; a token budget that pools or not"""
    g = torch.Generator().manual_seed(5000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g).item())       # noqa: E731
    vh = ri(1, 4)
    kv = ri(1, 2)
    heads = kv * (1, 2, 3, 7)[ri(0, 3)]
    arch = dict(vocab_size=320, hidden_size=128 * heads, intermediate_size=8 * ri(20, 150), num_hidden_layers=ri(1, 3), num_attention_heads=heads,
                num_key_value_heads=kv, rope_theta=(1e4, 1e6)[ri(0, 1)], mm_hidden_size=72 * vh, vision_token_num=(16, 72, 100000)[ri(0, 2)],
                vision=dict(image_size=(56, 62, 70, 84)[ri(0, 3)], patch_size=14, hidden_size=72 * vh, intermediate_size=4 * ri(20, 200),
                            num_hidden_layers=ri(2, 4), num_attention_heads=vh))
    T = ri(1, 9)
    toks = [ri(1, 319) for _ in range(ri(0, 5))] + [-200] + [ri(1, 319) for _ in range(ri(0, 8))]
    ids = torch.tensor([toks], dtype=torch.long)
    return dict(seed=seed, arch=arch, T=[T], ids=ids, mask=torch.ones_like(ids).bool(), side=arch["vision"]["image_size"],
                what=f"arch {seed}: tower {vh} x 72, ffn {arch['vision']['intermediate_size']}, {arch['vision']['num_hidden_layers']} layers, side "
                     f"{arch['vision']['image_size']} | decoder {heads} q / {kv} kv x 128, ffn {arch['intermediate_size']}, {arch['num_hidden_layers']} layers | "
                     f"T={T} budget={arch['vision_token_num']} prompt={len(toks)}")


def arch_config(s: dict):
    from videoitg_amd.config import ITGConfig
    return ITGConfig.from_dict(s["arch"])


def arch_images(s: dict):
    from videoitg_amd import synth
    return [synth.synth_pixel_values(t, s["side"], seed=500 + 7 * s["seed"] + i) for i, t in enumerate(s["T"])]
