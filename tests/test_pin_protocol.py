"""The small device->host read protocol (PinRead, pandelos_amd/csrc/pdl_common.h) on plain memory — no GPU.

A tiny kernel stores the words of up to 8 segments straight into pinned host memory, then a position-weighted checksum, then
an epoch flag; the host spins until the flag shows this read's epoch AND the words in place add up to the checksum (the words
travel over several paths: the flag alone can overtake the last of them).  Every counter read that decides "repeat the
pass" goes through it.  Here the host side (pdl_pin_arrived) is played against buffers in every state a read can be seen in."""
import ctypes as C

import numpy as np
import pytest

from pandelos_amd import _lib


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _segs(lengths, starts):
    return np.array(starts, np.uint32), np.array(lengths, np.uint32)


def _arrived(lib, pin, dst, words, flag_word, epoch):
    return lib.pdl_pin_arrived(pin.ctypes.data, dst.ctypes.data, words.ctypes.data, len(dst), flag_word, epoch)


def _land(lib, pin, payload, dst, words, flag_word, epoch, upto=None):
    """What the kernel does, in its order: words, then the checksum, then the flag (`upto` words only: the rest is still under way)."""
    at = 0
    for s in range(len(dst)):
        for i in range(int(words[s])):
            if upto is None or at < upto:
                pin[dst[s] + i] = payload[at]
            at += 1
    pin[flag_word + 1] = lib.pdl_pin_checksum(payload.ctypes.data, dst.ctypes.data, words.ctypes.data, len(dst))
    pin[flag_word] = epoch


def test_a_complete_read_arrives_and_an_old_epoch_does_not(lib):
    rng = np.random.default_rng(1)
    pin = np.zeros(4096, np.uint32)
    dst, words = _segs([12, 200, 3], [0, 16, 400])
    payload = rng.integers(0, 2 ** 32, int(words.sum()), dtype=np.uint32)
    flag = 4000
    assert _arrived(lib, pin, dst, words, flag, 1) == 0                     # nothing yet
    _land(lib, pin, payload, dst, words, flag, 1)
    assert _arrived(lib, pin, dst, words, flag, 1) == 1
    assert _arrived(lib, pin, dst, words, flag, 2) == 0                     # the NEXT read must not take the previous flag for its own


def test_flag_before_payload_is_not_an_arrival(lib):
    """The flag and the checksum are up but a word is still on its way: whichever word it is, the read has not arrived."""
    rng = np.random.default_rng(2)
    dst, words = _segs([8, 8, 40], [4, 64, 128])
    flag = 1000
    n = int(words.sum())
    for late in range(n):
        pin = rng.integers(0, 2 ** 32, 1024, dtype=np.uint32)              # stale content of earlier reads everywhere
        payload = rng.integers(1, 2 ** 32, n, dtype=np.uint32)
        stale = pin.copy()
        _land(lib, pin, payload, dst, words, flag, 7)
        seg = int(np.searchsorted(np.cumsum(words), late, side="right"))
        pos = int(dst[seg]) + late - int(np.cumsum(words)[seg] - words[seg])
        if stale[pos] == pin[pos]:
            continue
        pin[pos] = stale[pos]                                              # that one word has not landed yet
        assert _arrived(lib, pin, dst, words, flag, 7) == 0, f"word {late} missing and the read counts as arrived"
        pin[pos] = payload[late]
        assert _arrived(lib, pin, dst, words, flag, 7) == 1


def test_two_late_words_at_the_same_index_of_different_segments_do_not_cancel(lib):
    """With weights that restart in every segment (2 i + 1) a word late by +d in one segment and by -d at the same index of
    another cancelled; the weights follow the position in the whole buffer now."""
    dst, words = _segs([16, 16], [0, 512])
    flag = 2000
    payload = np.arange(100, 132, dtype=np.uint32)
    pin = np.zeros(2048 + 16, np.uint32)
    _land(lib, pin, payload, dst, words, flag, 3)
    assert _arrived(lib, pin, dst, words, flag, 3) == 1
    for i in range(16):
        for d in (1, 5, 12345):
            bad = pin.copy()
            bad[0 + i] += np.uint32(d)                                     # stale values that differ from the right ones by +d and -d
            bad[512 + i] -= np.uint32(d)
            assert _arrived(lib, bad, dst, words, flag, 3) == 0


def test_identical_payloads_of_consecutive_reads_are_told_apart_by_the_epoch(lib):
    dst, words = _segs([32], [0])
    flag = 100
    payload = np.full(32, 42, np.uint32)
    pin = np.zeros(256, np.uint32)
    _land(lib, pin, payload, dst, words, flag, 10)
    assert _arrived(lib, pin, dst, words, flag, 10) == 1
    assert _arrived(lib, pin, dst, words, flag, 11) == 0                    # same words, same checksum — but not this read's flag
    _land(lib, pin, payload, dst, words, flag, 11)
    assert _arrived(lib, pin, dst, words, flag, 11) == 1


def test_partial_landing_in_kernel_order(lib):
    """Words land in order, the checksum and flag last: at no point before the end does the host see an arrival."""
    rng = np.random.default_rng(5)
    dst, words = _segs([5, 7, 9], [8, 40, 80])
    flag = 500
    payload = rng.integers(1, 2 ** 32, int(words.sum()), dtype=np.uint32)
    for upto in range(int(words.sum())):
        pin = np.zeros(600, np.uint32)
        pin[flag] = 8                                                      # previous read's flag
        _land(lib, pin, payload, dst, words, flag, 9, upto=upto)
        assert _arrived(lib, pin, dst, words, flag, 9) == 0 or np.all(payload[upto:] == 0)
