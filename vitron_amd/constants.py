"""Sentinel ids and special tokens -- values of the reference's vitron/constants.py (part of the prompt format)."""
IGNORE_INDEX = -100            # vitron/constants.py:7
IMAGE_TOKEN_INDEX = -200       # :9
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
DEFAULT_VIDEO_TOKEN = "<video>"
OBJS_TOKEN_INDEX = -300        # :24
DEFAULT_OBJS_TOKEN = "<objs>"
MAX_IMAGE_LENGTH = 16          # :32
MAX_VIDEO_LENGTH = 1           # :33
