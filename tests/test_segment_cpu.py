"""hash.CRC32C (internal/hash/crc32c.go) as exported by the library: no GPU needed."""
import numpy as np

from tests.segfile import crc32c_py


def test_crc32c_known_answers():
    import vecgo_amd as vg
    assert vg.crc32c(b"123456789") == 0xE3069283           # the CRC-32C check value
    assert vg.crc32c(b"") == 0
    assert vg.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert vg.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of ones
    assert vg.crc32c(bytes(range(32))) == 0x46DD794E        # RFC 3720 B.4: 0x00..0x1F


def test_crc32c_matches_bitwise_reference_on_random_lengths():
    import vecgo_amd as vg
    rng = np.random.default_rng(1)
    for n in [1, 7, 8, 9, 63, 64, 65, 1000, 4099]:
        data = rng.integers(0, 256, n).astype(np.uint8).tobytes()
        assert vg.crc32c(data) == crc32c_py(data), n
