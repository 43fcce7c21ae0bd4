"""Token / label constants of the frame-selector path.

Values are fixed by the reference's checkpoints and call sites (/root/reference/eagle/constants.py:9-15: the `<image>`
placeholder is tokenised to -200, ignored label positions are -100); only the names the selector path uses are kept.
"""
IGNORE_INDEX, IMAGE_TOKEN_INDEX = -100, -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN = "<im_patch>", "<im_start>", "<im_end>"
