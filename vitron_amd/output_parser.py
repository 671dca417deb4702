"""What happens to the generated text right after `model.generate` in the reference's chat loop: the reply
is split into the visible answer and the routing fields that pick a task back-end. Mirror of `parse_model_output` and its four
helpers in the reference's app.py:345-395 (call site app.py:572-578); same return values for any string, checked against the
reference's own functions in tests/golden/output_parser.json.

The model marks up its reply with three kinds of tagged spans:
    <module>X</module>                 one letter naming the back-end (image generation, segmentation, video editing, ...)
    <instruction>k: v</instruction>    any number of them; the part after the LAST ':' of each is the argument
    <region>...</region>               an optional box / sketch reference
and the visible answer is the reply with every "<tag>...<tag>" stretch cut out (ANY pair of tags on one line, whatever their
names -- that is what the reference's pattern does, and callers depend on it to drop <SP>...</SP> as well).
"""
from __future__ import annotations

import re
from typing import List, NamedTuple, Optional

# spans never cross a newline ('.' without DOTALL), and the shortest span wins: both follow the reference's patterns
_SPAN = {name: re.compile(f"<{name}>(.*?)</{name}>") for name in ("module", "instruction", "region")}
_ANY_TAG_PAIR = re.compile(r"<[^>]+>(.*?)<[^>]+>")


class ParsedOutput(NamedTuple):
    output: str
    module: Optional[str]
    instruction: Optional[List[str]]
    region: Optional[str]


def _first(name: str, text: str) -> Optional[str]:
    m = _SPAN[name].search(text)
    return m.group(1) if m else None


def parse_model_output(model_output: str) -> ParsedOutput:
    """(visible answer, module or None, list of instruction arguments or None, region or None); unpacks like the
    reference's 4-tuple."""
    args = [body.rsplit(":", 1)[-1].strip() for body in _SPAN["instruction"].findall(model_output)]
    return ParsedOutput(_ANY_TAG_PAIR.sub("", model_output), _first("module", model_output), args or None,
                        _first("region", model_output))
