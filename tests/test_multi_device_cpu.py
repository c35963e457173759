"""How one deflate call is spread over several GPUs inside the library: the partition of the chunks and the scan that
places them, through the C ABI (these two entry points need no GPU).  The bytes are checked on the GPU
(test_gpu_multi_device.py)."""
import random

import pytest

import zlib_amd


@pytest.mark.parametrize("n_devices", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("total", [1, 2, 3, 7, 8, 9, 63, 64, 65, 1000, 8192, 8193])
def test_partition_is_contiguous_and_chunk_c_is_entry_c(n_devices, total):
    parts = [zlib_amd.multi_partition(total, n_devices, i) for i in range(n_devices)]
    per = parts[0][2]
    assert per == -(-total // n_devices) and all(p[2] == per for p in parts)
    at = 0
    seen_short = False
    for i, (first, count, _) in enumerate(parts):
        assert first == min(at, total) and count <= per
        if count:
            # a table of n_devices * per entries, device after device: this device's chunks start at its own block
            assert first == i * per
            assert not seen_short, "only the last device that has chunks may have fewer than the others"
        if count < per:
            seen_short = True
        at += count
    assert at == total


@pytest.mark.parametrize("n_devices", [2, 4, 8])
def test_scan_of_the_padded_table_places_every_chunk(n_devices):
    rng = random.Random(n_devices)
    for total in (1, 5, 64, 1001):
        lengths = [rng.randrange(1, 1 << 21) for _ in range(total)]
        per = zlib_amd.multi_partition(total, n_devices, 0)[2]
        table = [0] * (n_devices * per)
        for i in range(n_devices):
            first, count, _ = zlib_amd.multi_partition(total, n_devices, i)
            # what device i contributes to the all-gather: its chunks' lengths, then chunks of no bytes
            table[i * per:i * per + count] = lengths[first:first + count]
        offs = zlib_amd.multi_offsets_host(table, base=7)
        want = [7]
        for v in lengths:
            want.append(want[-1] + v)
        assert offs[:total + 1] == want
        assert all(o == want[-1] for o in offs[total:])          # the padding repeats the total
        for i in range(n_devices):
            first, count, _ = zlib_amd.multi_partition(total, n_devices, i)
            if count:
                assert offs[i * per] == want[first]              # where device i's bytes belong


def test_bad_arguments_are_refused():
    with pytest.raises(zlib_amd.ZamdError):
        zlib_amd.multi_partition(10, 0, 0)
    with pytest.raises(zlib_amd.ZamdError):
        zlib_amd.multi_partition(10, 2, 2)
