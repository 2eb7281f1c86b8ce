"""The rule that picks the 256-row tile form of the split-precision GEMM (encoder.hip x3_big_form, exported as
mvdb_encoder_gemm_tile_form): a host-side restatement and the shapes DESIGN.md quotes.  No GPU needed."""
import pytest

from minivectordb_amd import _native as native


def _rule(tokens, n, cus):
    bn = 256 if n % 256 == 0 else 192 if n % 192 == 0 else 0
    if not bn or tokens <= 0:
        return 0
    tiles = (n // bn) * ((tokens + 255) // 256)
    rounds = (tiles + cus - 1) // cus
    return bn if tiles >= cus and (tiles >= 4 * cus or tiles * 100 >= rounds * cus * 85) else 0


@pytest.mark.parametrize("tokens,n,want", [
    # e5-large shape (H 1024, FFN 4096) on 256 CUs
    (8192, 3072, 0),      # 256 x 32 tokens, QKV: 384 tiles = 1.5 rounds
    (8192, 1024, 0),      # 128 tiles: half a round
    (8192, 4096, 256),    # FFN1: 512 tiles = two full rounds
    (16384, 1024, 256),   # 256 x 64 tokens: exactly one round
    (16384, 3072, 256),
    (131072, 4096, 256),
    # e5-small shape (H 384, FFN 1536): every N is a multiple of 192, none of 256 but 1536
    (131072, 384, 192), (131072, 1152, 192), (131072, 1536, 256),
    (32768, 384, 192),    # 256 x 128 slots, full: 256 tiles = one round
    (20592, 384, 0),      # the same batch ragged: 162 tiles
    (20592, 1152, 192),   # 486 tiles: second round 90 % full
    (8192, 1536, 0),      # the S = 32 batch stays on the 64- / 128-row forms
    (8192, 1152, 0), (8192, 384, 0),
    # widths without a 256-row form
    (131072, 96, 0), (131072, 160, 0), (131072, 1280, 256), (131072, 320, 0),
])
def test_tile_form_of_quoted_shapes(tokens, n, want):
    assert native.encoder_gemm_tile_form(tokens, n, 256) == want == _rule(tokens, n, 256)


def test_tile_form_matches_restatement_on_a_grid():
    for cus in (64, 256, 304):
        for n in (96, 192, 256, 384, 768, 1024, 1152, 1536, 3072, 4096):
            for tokens in list(range(1, 4000, 173)) + [8192, 10006, 16384, 20592, 32768, 40118, 65536, 82731, 131072, 1 << 22]:
                assert native.encoder_gemm_tile_form(tokens, n, cus) == _rule(tokens, n, cus), (tokens, n, cus)
    assert native.encoder_gemm_tile_form(0, 1024, 256) == 0
    assert native.encoder_gemm_tile_form(4096, 0, 256) == 0
    assert native.encoder_gemm_tile_form(4096, 1024, 0) == 0


def _planes(tokens, n, k, cus):
    """Restatement of encoder.hip: x3_splitk_parts (default switches)."""
    tiles = ((tokens + 63) // 64) * ((n + 127) // 128)
    parts = min(8, k // 32 // 4, cus // max(tiles, 1))
    if parts < 2 or parts * tokens * n > cus * 64 * 128 + 64 * 1024:
        return 0
    return {5: 4, 7: 6}.get(parts, parts)


@pytest.mark.parametrize("tokens,n,k,want", [
    # one long sentence, e5-small shape (H 384, FFN 1536) on 256 CUs
    (256, 384, 1536, 8),    # FFN2: 48 K-steps, 12 tiles
    (256, 384, 384, 3),     # attention output projection: 12 K-steps -> three planes of four
    (512, 384, 1536, 8), (1024, 384, 1536, 4), (2048, 384, 1536, 2), (8192, 384, 1536, 0),   # the 256 x 32 batch is not split
    # bge-m3 / e5-large shape (H 1024, FFN 4096)
    (129, 1024, 4096, 8), (256, 1024, 4096, 8), (384, 1024, 4096, 4), (512, 1024, 4096, 4),
    (129, 3072, 1024, 3), (129, 4096, 1024, 2),   # QKV / FFN1 (the library uses them from three planes on)
    (1024, 1024, 4096, 2), (2048, 1024, 4096, 0),
])
def test_splitk_planes_of_quoted_shapes(tokens, n, k, want):
    assert native.encoder_splitk_planes(tokens, n, k, 256) == want == _planes(tokens, n, k, 256)


def test_splitk_planes_match_restatement_on_a_grid():
    for cus in (64, 256, 304):
        for n, k in ((384, 384), (384, 1536), (1152, 384), (1536, 384), (1024, 1024), (1024, 4096), (3072, 1024), (4096, 1024), (768, 3072)):
            for tokens in list(range(1, 3000, 97)) + [4096, 8192, 131072]:
                assert native.encoder_splitk_planes(tokens, n, k, cus) == _planes(tokens, n, k, cus), (tokens, n, k, cus)
    assert native.encoder_splitk_planes(0, 384, 384, 256) == 0
