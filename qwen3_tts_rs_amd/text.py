"""Text → token ids for the CLI / e2e bench. The reference wraps the HF `tokenizers` crate (src/tokenizer/text.rs:62-217:
`tokenizer.json`, or vocab.json + merges.txt); here the same `tokenizer.json` is loaded through the `tokenizers` Python
package when a model directory has one. Without a tokenizer (synthetic checkpoints: there is no network to fetch the
Qwen2 vocabulary) a deterministic stand-in maps each word piece to an id below the text vocabulary, so the CLI and the
benchmark exercise prompts of realistic length; it is NOT the Qwen2 BPE and is labelled as such in the outputs."""
import os
import re
import zlib
from typing import List, Optional


class TextTokenizer:
    def __init__(self, tokenizer_json: Optional[str] = None, vocab_limit: int = 151643):
        self.kind = "synthetic-wordpiece"
        self._tok = None
        self._limit = vocab_limit
        if tokenizer_json and os.path.exists(tokenizer_json):
            from tokenizers import Tokenizer
            self._tok = Tokenizer.from_file(tokenizer_json)
            self.kind = "tokenizer.json"

    @classmethod
    def from_pretrained(cls, model_dir: Optional[str], tokenizer_dir: Optional[str] = None, allow_stand_in: bool = False) -> "TextTokenizer":
        """<tokenizer_dir>/tokenizer.json, else model_dir/tokenizer.json, else model_dir/../tokenizer/tokenizer.json
        (generate_audio.rs:48-50). A real checkpoint without a tokenizer is an ERROR, as in the reference
        (TextTokenizer::from_pretrained, text.rs:62-110): synthesizing from stand-in ids would silently produce
        meaningless audio. The stand-in is handed out only when the caller asks for it (`allow_stand_in`: synthetic
        checkpoints, or no model directory at all)."""
        cands = []
        if tokenizer_dir:
            cands.append(os.path.join(tokenizer_dir, "tokenizer.json"))
        if model_dir:
            cands += [os.path.join(model_dir, "tokenizer.json"), os.path.join(os.path.dirname(os.path.abspath(model_dir)), "tokenizer", "tokenizer.json")]
        for c in cands:
            if os.path.exists(c):
                return cls(c)
        if model_dir is not None and not allow_stand_in:
            raise FileNotFoundError("Failed to load tokenizer: no tokenizer.json in " + ", ".join(os.path.dirname(c) for c in cands) +
                                    " (pass --tokenizer-dir, or --token-ids to bypass the tokenizer)")
        return cls(None)

    def encode(self, text: str) -> List[int]:
        if self._tok is not None:
            return list(self._tok.encode(text).ids)
        ids = []
        for piece in re.findall(r"\w+|[^\w\s]", text, flags=re.UNICODE):
            # ~1.3 ids per word like a BPE on English: words longer than 6 characters split in two
            parts = [piece] if len(piece) <= 6 else [piece[: len(piece) // 2], piece[len(piece) // 2:]]
            ids += [zlib.crc32(p.encode("utf-8")) % self._limit for p in parts]
        return ids
