"""Bank arithmetic of the prefill attention kernel's K / V ring (exl3_attn_prefill.hip, attn_prefill_w64_kernel<HD>), restated on the host: the tiles are
unpadded rows of 16-byte chunks, position p of row r holding chunk p ^ f(r); these tests pin the properties the kernel's comments claim, with the lane
groups and bank rule of /opt/skills/guides/MI355X_MICROARCH.md (LDS table): ds_read_b128 is served in four groups of 16 lanes, ds_read_b64_tr_b16 in
two groups of 32, bank = (byte address / 4) mod 64, and only lanes of one group conflict.  (GPU side: SQ_LDS_BANK_CONFLICT = 0.9 % of SQ_LDS_IDX_ACTIVE,
profiles/r04_attn_prefill_ablations.txt.)"""
import itertools
import pytest

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def kswz(hd, r):
    return (r & 15) if hd == 128 else ((r >> 1) & 7)


def vswz(hd, r):
    return (((r & 3) << 2) | ((r >> 2) & 3)) if hd == 128 else (((r >> 1) & 1) << 2)


def banks(byte_addr, nbytes):
    return {((byte_addr + 4 * i) // 4) % 64 for i in range(nbytes // 4)}


@pytest.mark.parametrize("hd", [128, 64])
def test_swizzles_are_permutations_of_a_row(hd):
    ch = hd // 8
    for r in range(64):
        assert sorted(p ^ kswz(hd, r) for p in range(ch)) == list(range(ch))
        assert sorted(p ^ vswz(hd, r) for p in range(ch)) == list(range(ch))


@pytest.mark.parametrize("hd", [128, 64])
def test_k_fragment_reads_touch_every_bank_once_per_lane_group(hd):
    """A operand of S^T: lane (n = l % 32, h = l / 32) reads 16 bytes of row 32 kb + n, chunk 2 ks + h."""
    for ks, kb in itertools.product(range(hd // 16), range(2)):
        for grp in B128_GROUPS:
            seen = set()
            for l in grp:
                n, h = l & 31, l >> 5
                r = 32 * kb + n
                addr = r * hd * 2 + (((2 * ks + h) ^ kswz(hd, r)) * 16)
                b = banks(addr, 16)
                assert not (b & seen), (hd, ks, kb, l)
                seen |= b
            assert len(seen) == 64


@pytest.mark.parametrize("hd", [128, 64])
def test_v_transpose_reads_touch_every_bank_once_per_lane_group(hd):
    """B operand of P V: lane (G = l / 16, c = l % 16) hands in 8 bytes of row 16 kst + 4 (G >> 1) + (c >> 2) (+ 8 for the second read), halves
    32 nb + 16 (G & 1) + 4 (c & 3) ..: chunk 4 nb + 2 (G & 1) + ((c & 3) >> 1), half (c & 1) of it."""
    for kst, nb, second in itertools.product(range(4), range(hd // 32), range(2)):
        for half_wave in range(2):
            seen = set()
            for l in range(32 * half_wave, 32 * half_wave + 32):
                G, c = l >> 4, l & 15
                r = 16 * kst + 4 * (G >> 1) + (c >> 2) + 8 * second
                chunk = 4 * nb + 2 * (G & 1) + ((c & 3) >> 1)
                addr = r * hd * 2 + ((chunk ^ vswz(hd, r)) * 16) + 8 * (c & 1)
                b = banks(addr, 8)
                assert not (b & seen), (hd, kst, nb, second, l)
                seen |= b
            assert len(seen) == 64


@pytest.mark.parametrize("hd", [128, 64])
def test_reader_offsets_are_slice_independent(hd):
    """The kernel keeps ONE pointer per (k-step) / (output block) and puts key block, P V k-step and ring slot into the read's immediate offset: the
    swizzle of row r + 32 (K) and of row r + 16 kst (V) must equal that of row r."""
    for r in range(32):
        assert kswz(hd, r + 32) == kswz(hd, r)
    for r, kst in itertools.product(range(16), range(4)):
        assert vswz(hd, r + 16 * kst) == vswz(hd, r)
