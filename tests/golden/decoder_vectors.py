"""Hand-derived known answers for `Decoder::decode_to_str_without_replacement` where the reference's own tests pin
nothing ("parity unpinned"): the UTF-16 surrogate accounting and the UTF-8 un-read rule.

Source: the WHATWG Encoding Standard's "utf-16 decoder" / "utf-8 decoder" algorithms, and for what a streaming
decoder must do differently the crate's documented contract as restated in SURVEY.md 3.3:
  * a low surrogate without a high one: Malformed, the unit consumed;
  * a high surrogate followed IN THE SAME CALL by a whole unit that is not a low surrogate: Malformed, only the high
    surrogate consumed (the following unit is decoded by the next call);
  * a high surrogate that is the LAST whole unit of a call stays pending; the next call's first unit decides:
    low -> the pair; high -> Malformed, that unit consumed, the new high surrogate pending; anything else ->
    Malformed, that unit consumed too, and its character is the FIRST OUTPUT of the call after that (pending_bmp);
  * an odd byte at the end of a call stays pending (the unit grid is the stream's, not the call's);
  * UTF-8: a byte that cannot continue the sequence is NOT consumed by the Malformed call (it starts the next
    call); E0 / ED / F0 / F4 narrow the range of the second byte; C0, C1, F5..FF and stray continuation bytes are
    one-byte errors.
Each vector was written down from these rules BEFORE it was run against any implementation; it is checked against the
oracle's decoders (oracle/sxo.c) and the product's (sx_codec_core.hpp compiled for the host — the device kernels are
compiled from the same source) in tests/test_decoder_vectors.py, and end to end on the GPU through finding positions.

Format: (name, encoding, [call, ...]); call = (input bytes as hex, last flag, [(result, read, written), ...]) — the
steps the loop `loop { r = decode(rest); rest = rest[read..]; if r is InputEmpty { break } }` goes through
(src/finding_collection.rs:134-143,292-325: after Malformed the decoder is called again, even on empty input).
result: E = InputEmpty, M = Malformed."""

H, H2, L, A, B = "3dd8", "3cd8", "00de", "4100", "4200"      # UTF-16LE units: U+D83D, U+D83C (high), U+DE00 (low), 'A', 'B'


def be(hexunits):
    """the same units, big endian"""
    return "".join(hexunits[i + 2:i + 4] + hexunits[i:i + 2] for i in range(0, len(hexunits), 4))


UTF16LE = [
    ("bmp", [(A + B, False, [("E", 4, 2)])]),
    ("pair", [(H + L, False, [("E", 4, 4)])]),
    ("lone low", [(L + A, False, [("M", 2, 0), ("E", 2, 1)])]),
    ("high then bmp in one call", [(H + A, False, [("M", 2, 0), ("E", 2, 1)])]),
    ("high high low in one call", [(H + H2 + L, False, [("M", 2, 0), ("E", 4, 4)])]),
    ("high as last unit stays pending, low completes it", [(H, False, [("E", 2, 0)]), (L, False, [("E", 2, 4)])]),
    ("pending high then bmp: both consumed, the bmp char opens the call after", [
        (H, False, [("E", 2, 0)]), (A + B, False, [("M", 2, 0), ("E", 2, 2)])]),
    ("pending high then bmp at the very end: an empty call delivers it", [
        (H, False, [("E", 2, 0)]), (A, False, [("M", 2, 0), ("E", 0, 1)])]),
    ("pending high then high then low", [(H, False, [("E", 2, 0)]), (H2 + L, False, [("M", 2, 0), ("E", 2, 4)])]),
    ("pending high then lone low is the pair even across three calls", [
        (A + H, False, [("E", 4, 1)]), (L + B, False, [("E", 4, 5)])]),
    ("odd byte pending", [("41", False, [("E", 1, 0)]), ("00" + B, False, [("E", 3, 2)])]),
    ("high + half a unit", [(H + "41", False, [("E", 3, 0)]), ("00", False, [("M", 1, 0), ("E", 0, 1)])]),
    ("pending high at the end of the input", [(A + H, True, [("M", 4, 1), ("E", 0, 0)])]),
    ("pending odd byte at the end of the input", [(A + "41", True, [("M", 3, 1), ("E", 0, 0)])]),
    ("low low", [(L + L, False, [("M", 2, 0), ("M", 2, 0), ("E", 0, 0)])]),
]

UTF8 = [
    ("ascii", [("4142", False, [("E", 2, 2)])]),
    ("c0 is a one-byte error", [("c0af41", False, [("M", 1, 0), ("M", 1, 0), ("E", 1, 1)])]),
    ("lead + ascii: the ascii byte is not consumed", [("c241", False, [("M", 1, 0), ("E", 1, 1)])]),
    ("e0 narrows the second byte to a0..bf", [("e09f80", False, [("M", 1, 0), ("M", 1, 0), ("M", 1, 0), ("E", 0, 0)])]),
    ("e0 a0 80 is U+0800", [("e0a080", False, [("E", 3, 3)])]),
    ("ed narrows to 80..9f (no surrogates)", [("eda080", False, [("M", 1, 0), ("M", 1, 0), ("M", 1, 0), ("E", 0, 0)])]),
    ("f0 narrows to 90..bf", [("f08f8080", False, [("M", 1, 0), ("M", 1, 0), ("M", 1, 0), ("M", 1, 0), ("E", 0, 0)])]),
    ("f4 narrows to 80..8f", [("f4908080", False, [("M", 1, 0), ("M", 1, 0), ("M", 1, 0), ("M", 1, 0), ("E", 0, 0)])]),
    ("f5 and ff", [("f5ff41", False, [("M", 1, 0), ("M", 1, 0), ("E", 1, 1)])]),
    ("two good bytes then a bad third: three bytes malformed, the bad byte stays", [("e282" + "41", False, [("M", 2, 0), ("E", 1, 1)])]),
    ("three good bytes of four then ascii", [("f09f98" + "41", False, [("M", 3, 0), ("E", 1, 1)])]),
    ("a sequence split over two calls", [("e2", False, [("E", 1, 0)]), ("82ac41", False, [("E", 3, 4)])]),
    ("a bad continuation as the FIRST byte of a call: nothing read, nothing written", [
        ("e2", False, [("E", 1, 0)]), ("41", False, [("M", 0, 0), ("E", 1, 1)])]),
    ("truncated at the end of the input", [("41e282", True, [("M", 3, 1), ("E", 0, 0)])]),
    ("stray continuation bytes", [("80bf41", False, [("M", 1, 0), ("M", 1, 0), ("E", 1, 1)])]),
]

# gb18030 / GBK — the WHATWG "gb18030 decoder" (https://encoding.spec.whatwg.org/#gb18030-decoder): first / second / third,
# "prepend ... to ioQueue" = the bytes are read again.  What a streaming decoder does with bytes it must read again but received in
# an EARLIER call is this project's choice (encoding_rs is not vendored: unpinned): they are decoded in front of the next call's
# input; read counts bytes of the call's own input only.
GB = [
    ("two bytes", [("d6d0", False, [("E", 2, 3)])]),                                     # U+4E2D
    ("four bytes, BMP range", [("81308130", False, [("E", 4, 2)])]),                     # pointer 0 -> U+0080
    ("four bytes, astral", [("90308130", False, [("E", 4, 4)])]),                        # pointer 189000 -> U+10000
    ("0x80 is the euro sign", [("80", False, [("E", 1, 3)])]),
    ("0xFF is an error of its own", [("ff41", False, [("M", 1, 0), ("E", 1, 1)])]),
    ("lead + ASCII byte that is no trail: the byte is read again", [("a17f", False, [("M", 1, 0), ("E", 1, 1)])]),
    ("lead + 0xFF: both consumed", [("a1ff", False, [("M", 2, 0), ("E", 0, 0)])]),
    # (round 3, ADVICE: the digit / the third byte stay consumed — the crate keeps them pending — and are decoded in front of
    # the next call's input; only the byte that broke the token is read again)
    ("lead digit, then no lead: that byte is read again, the digit comes out in front of it", [("81307841", False, [("M", 2, 0), ("E", 2, 3)])]),
    ("lead digit lead, then no digit: that byte is read again, digit and third byte are decoded in front of it",
     [("81308178", False, [("M", 3, 0), ("E", 1, 4)])]),   # '0' + U+4E41
    ("pointer between the BMP ranges and the astral planes: error, four bytes consumed", [("8431a530", False, [("M", 4, 0), ("E", 0, 0)])]),
    ("the token over three calls, then the error: what earlier calls consumed is decoded in front of the next call",
     [("81", False, [("E", 1, 0)]), ("3081", False, [("E", 2, 0)]), ("78", False, [("M", 0, 0), ("E", 1, 4)])]),
    ("pending bytes at the end of the stream", [("8130", True, [("M", 2, 0), ("E", 0, 0)])]),
]

# ISO-2022-JP (WHATWG "ISO-2022-JP decoder"; round 3).  ESC ( B = ASCII, ESC ( J = JIS X 0201 Roman, ESC ( I = half-width katakana,
# ESC $ @ / ESC $ B = JIS X 0208 (two bytes per character; 0x2422 = U+3042).  What a streaming decoder adds to the algorithm: a byte
# "restored to the stream" is read again by the next call (`read` stops in front of it); the `$` / `(` of an escape sequence that
# fails at its third byte was already consumed — its character is the FIRST OUTPUT of the call after the error (as the crate's
# pending_prepended; like pending_bmp in UTF-16).  Unpinned like every legacy decoder.
ISO2022JP = [
    ("plain ASCII", [("4142", False, [("E", 2, 2)])]),
    ("to JIS X 0208 and back", [("1b244224221b284241", False, [("E", 9, 4)])]),
    ("Roman: 5C is the yen sign, 7E the overline", [("1b284a5c7e41", False, [("E", 6, 6)])]),
    ("katakana", [("1b2849215f", False, [("E", 5, 6)])]),
    ("katakana: 0x60 is an error, the set stays", [("1b28496041", False, [("M", 4, 0), ("E", 1, 3)])]),
    ("shift out is an error in ASCII", [("0e41", False, [("M", 1, 0), ("E", 1, 1)])]),
    ("a byte >= 0x80 is an error", [("a441", False, [("M", 1, 0), ("E", 1, 1)])]),
    ("two escape sequences in a row: the second is the error", [("1b28421b284a7e", False, [("M", 6, 0), ("E", 1, 3)])]),
    ("ESC and no escape sequence: the ESC is the error, the byte is read again", [("1b41", False, [("M", 1, 0), ("E", 1, 1)])]),
    ("ESC ( and a wrong third byte: that byte is read again, the ( comes out in front of it",
     [("1b2841", False, [("M", 2, 0), ("E", 1, 2)])]),
    ("ESC $ and a wrong third byte", [("1b2441", False, [("M", 2, 0), ("E", 1, 2)])]),
    ("an escape sequence over three calls", [("1b", False, [("E", 1, 0)]), ("24", False, [("E", 1, 0)]), ("42", False, [("E", 1, 0)]),
                                             ("2422", False, [("E", 2, 3)])]),
    ("a broken one over three calls: nothing of the last call is read, the ( belongs to the call after",
     [("1b", False, [("E", 1, 0)]), ("28", False, [("E", 1, 0)]), ("41", False, [("M", 0, 0), ("E", 1, 2)])]),
    ("ESC pending, then no escape sequence", [("1b", False, [("E", 1, 0)]), ("41", False, [("M", 0, 0), ("E", 1, 1)])]),
    ("ESC instead of a second byte: the first byte is the error, the escape sequence goes on",
     [("1b2442241b284241", False, [("M", 5, 0), ("E", 3, 1)])]),
    ("a second byte out of range: both consumed, the set stays", [("1b2442242041", False, [("M", 5, 0), ("E", 1, 0)])]),
    ("a character over two calls", [("1b244224", False, [("E", 4, 0)]), ("22", False, [("E", 1, 3)])]),
    ("first byte pending at the end of the stream", [("1b244224", True, [("M", 4, 0), ("E", 0, 0)])]),
    ("ESC ( at the end of the stream: the ( is still decoded", [("1b28", True, [("M", 2, 0), ("E", 0, 1)])]),
    ("ESC at the end of the stream", [("1b", True, [("M", 1, 0), ("E", 0, 0)])]),
]

VECTORS = ([(n, "gb18030", c) for n, c in GB] + [(n, "iso-2022-jp", c) for n, c in ISO2022JP] + [(n, "gbk", c) for n, c in GB]
           + [(n, "utf-16le", c) for n, c in UTF16LE]
           + [(n, "utf-16be", [(be(x) if len(x) % 4 == 0 else None, last, steps) for x, last, steps in c]) for n, c in UTF16LE
              if all(len(x) % 4 == 0 for x, _, _ in c)]
           + [(n, "utf-8", c) for n, c in UTF8])
