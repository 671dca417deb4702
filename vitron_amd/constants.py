"""Sentinel ids and special tokens -- values of the reference's vitron/constants.py (part of the prompt format)."""
IGNORE_INDEX = -100            # vitron/constants.py:7
IMAGE_TOKEN_INDEX = -200       # :9
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
IMAGE_PLACEHOLDER = "<image-placeholder>"
DEFAULT_VIDEO_TOKEN = "<video>"
DEFAULT_VIDEO_PATCH_TOKEN = "<im_patch>"   # :18 -- the SAME string as the image patch token: add_tokens adds one id for both
DEFAULT_VID_START_TOKEN = "<vid_start>"
DEFAULT_VID_END_TOKEN = "<vid_end>"
OBJS_TOKEN_INDEX = -300        # :24
DEFAULT_OBJS_TOKEN = "<objs>"
MAX_IMAGE_LENGTH = 16          # :32
MAX_VIDEO_LENGTH = 1           # :33
