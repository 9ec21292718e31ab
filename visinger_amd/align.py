"""Host-side integer bookkeeping next to the hot path: the note list -> frame alignment of the reference's
``utils/audio/align.py:58-104`` (``get_note2dur``).  It runs once per utterance at data-preparation time on a few dozen notes -- list
and index arithmetic, no tensor work -- so it stays on the host like the reference's; the per-frame tensors it produces
(``mel2phone``) are what ``expand_states`` / ``mel2token_to_dur`` (csrc/index_ops.hip) consume on the device.  Bit-exact against
the reference's own outputs: tests/golden/note2dur.json."""
import numpy as np


def frame_of(seconds, sample_rate, hop_size):
    """align.py:81-82 / 89-90: nearest frame of a time stamp (round half up)"""
    return int(seconds * sample_rate / hop_size + 0.5)


def merge_notes(midi_info, min_sil_duration=0):
    """align.py:60-68: close gaps shorter than `min_sil_duration` (the previous note is stretched to the next start) and fold a rest
    ("|") that follows a rest into it.  Rows are (bar, pos, pitch, duration, start, end, tempo, phones, token); like the reference the
    rows are edited in place and the merged list shares them."""
    merged = []
    for i, midi in enumerate(midi_info):
        if i > 0 and midi[4] - merged[-1][5] < min_sil_duration:
            merged[-1][5] = midi[4]
        if i > 0 and midi[8] == "|" and merged[-1][8] == "|":
            merged[-1][5] = midi[5]
        else:
            merged.append(midi)
    return merged


def get_note2dur(midi_info, hop_size, sample_rate, min_sil_duration=0):
    """utils/audio/align.py:58-104 -> (mel2phone, mel2note, duration, ph_list, merged notes): every frame gets the 1-based index of its
    phoneme and of its note; a syllable of two phonemes gives its first 3 frames to the onset, one of three its first and last 3 frames to
    onset and coda; `duration[i]` = frames of phoneme i + 1 (the histogram of utils/audio/align.py:105-129)."""
    notes = merge_notes(midi_info, min_sil_duration)
    last_frame = frame_of(notes[-1][5], sample_rate, hop_size)
    mel2phone = np.zeros([last_frame], dtype=int)
    mel2note = np.zeros([last_frame], dtype=int)
    ph_list = []
    i_ph = 0
    for i_note, midi in enumerate(notes):
        start, end = frame_of(midi[4], sample_rate, hop_size), frame_of(midi[5], sample_rate, hop_size)
        n = len(midi[7])
        if n == 1:
            mel2phone[start:end] = i_ph + 1
        elif n == 2:
            mel2phone[start:start + 3] = i_ph + 1
            mel2phone[start + 3:end] = i_ph + 2
        elif n == 3:
            mel2phone[start:start + 3] = i_ph + 1
            mel2phone[start + 3:end - 3] = i_ph + 2
            mel2phone[end - 3:end] = i_ph + 3
        if n in (1, 2, 3):
            i_ph += n
        ph_list.extend(midi[7])
        mel2note[start:end] = i_note + 1
    mel2phone[-1] = mel2phone[-2]
    mel2note[-1] = mel2note[-2]
    assert not np.any(mel2phone == 0) and not np.any(mel2note == 0), f"| mel2phone: {mel2phone}, mel2note: {mel2note}, midi_info: {midi_info}"
    # frames per phoneme: integer histogram over 1..T_ph (index 0 = padding, dropped)
    duration = np.bincount(mel2phone, minlength=len(ph_list) + 1)[1:len(ph_list) + 1]
    return mel2phone.tolist(), mel2note.tolist(), duration.tolist(), ph_list, notes
