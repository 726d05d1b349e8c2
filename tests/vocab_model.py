"""A SentencePiece model with the reference's id map, rebuilt from tests/golden/vocab_pieces.json (test infrastructure).

The reference's tokenizer file (checkpoints/bpe.model) is not in its tree; what it holds is vocab.txt, the id -> piece dump
of that model (12000 pieces: 0-2 control, 3-6 language tags, 7-8472 CJK characters, 8473-10200 upper-case pinyin with tone
digit, 10201.. BPE word pieces).  This builds a unigram model over exactly those pieces in exactly that order, so every
id a test sees is the id the reference's model assigns to that piece.  Segmentation where several piece sequences spell the
same text follows the scores given here (longer pieces preferred), which need not be the reference's BPE merges: tests
only rely on unambiguous cases (single CJK characters, pinyin syllables, punctuation) and on id-level properties.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def pieces():
    return json.load(open(os.path.join(HERE, "golden", "vocab_pieces.json"), encoding="utf-8"))


def write_model(path: str) -> str:
    from sentencepiece import sentencepiece_model_pb2 as pb
    m = pb.ModelProto()
    seen = set()
    for i, p in enumerate(pieces()):
        sp = m.pieces.add()
        if p in seen:
            # vocab.txt was dumped after a simplified -> traditional conversion of the pieces: 62 CJK characters appear twice.
            # The text form maps to the FIRST id; the later slot is kept (ids must not shift) as an unused placeholder.
            sp.piece, sp.type, sp.score = f"<dup:{i}>", pb.ModelProto.SentencePiece.UNUSED, 0.0
            continue
        seen.add(p)
        sp.piece = p
        if i == 0 or i == 1:
            sp.type, sp.score = pb.ModelProto.SentencePiece.CONTROL, 0.0
        elif i == 2:
            sp.type, sp.score = pb.ModelProto.SentencePiece.UNKNOWN, 0.0
        else:
            sp.type = pb.ModelProto.SentencePiece.NORMAL
            sp.score = -10.0 + 2.5 * min(len(p), 8) - 1e-4 * i     # one long piece beats the same text in short ones
    t = m.trainer_spec
    t.model_type = pb.TrainerSpec.UNIGRAM
    t.vocab_size = len(m.pieces)
    t.unk_id, t.bos_id, t.eos_id, t.pad_id = 2, 0, 1, -1
    t.unk_piece, t.bos_piece, t.eos_piece = "<unk>", "<s>", "</s>"
    n = m.normalizer_spec
    n.name = "identity"
    n.add_dummy_prefix = True
    n.remove_extra_whitespaces = True
    n.escape_whitespaces = True
    with open(path, "wb") as f:
        f.write(m.SerializeToString())
    return path
