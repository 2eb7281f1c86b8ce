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
