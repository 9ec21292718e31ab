"""Batched synthesis driver: what the reference does one utterance at a time in
``tasks/visinger.py:244-263`` (``test_step``) and ``inference/visinger.py:91-100``, done in length-bucketed batches on
the MI355X-native model, plus the reference's wav normalisation (``utils/audio/io.py:8-15``: peak-normalise, scale to
int16).  Host-side plumbing only."""
import numpy as np
import torch


def bucket_by_length(lengths, max_frames_per_batch, max_items_per_batch=256, keys=None):
    """Sort by length (descending) and cut batches under a padded-frame budget (the reference's batch_by_size idea,
    utils/commons/dataset_utils.py:181-191, without its dataset plumbing).  Returns lists of item indices.
    keys (optional, one per item): items of different key never share a batch (batches are cut where the key changes)."""
    order = sorted(range(len(lengths)), key=lambda i: ((keys[i] if keys is not None else 0), -int(lengths[i])))
    batches, cur = [], []
    for i in order:
        longest = int(lengths[cur[0]]) if cur else int(lengths[i])
        if cur and (longest * (len(cur) + 1) > max_frames_per_batch or len(cur) >= max_items_per_batch or
                    (keys is not None and keys[i] != keys[cur[0]])):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def collate(items, device):
    """items: dicts with int64 1-D arrays text_tokens, pitch_tokens, dur_tokens (T_ph) and mel2ph (T_mel); 0-padded."""
    Tph = max(len(it["text_tokens"]) for it in items)
    T = max(len(it["mel2ph"]) for it in items)
    out = {k: torch.zeros((len(items), Tph if k != "mel2ph" else T), dtype=torch.long)
           for k in ("text_tokens", "pitch_tokens", "dur_tokens", "mel2ph")}
    for b, it in enumerate(items):
        for k in out:
            v = torch.as_tensor(np.asarray(it[k]), dtype=torch.long)
            out[k][b, :len(v)] = v
    out["spk_id"] = torch.as_tensor([int(it.get("spk_id", 0)) for it in items], dtype=torch.long)
    return {k: v.to(device) for k, v in out.items()}


def to_int16(wav, norm=True):
    """utils/audio/io.py:8-15: optional peak normalisation, then * 32767 -> int16."""
    wav = np.asarray(wav)
    if not np.issubdtype(wav.dtype, np.floating):
        wav = wav.astype(np.float32)
    # (the arithmetic stays in the array's own float type, as in the reference: an fp64 waveform is scaled in fp64 -- the int16 truncation
    #  of the two differs in the last bit; pinned byte for byte by tests/golden/save_wav.npz)
    if norm and wav.size:
        peak = np.abs(wav).max()
        if peak > 0:
            wav = wav / peak
    return (wav * 32767).astype(np.int16)


def save_wav(wav, path, sr, norm=False):
    """utils/audio/io.py:8-12: write `wav` (float, nominally in [-1, 1]) as 16-bit PCM to path[:-4] + '.wav'."""
    from scipy.io import wavfile
    wavfile.write(path[:-4] + ".wav", sr, to_int16(wav, norm=norm))


class StreamRotation:
    """Consecutive batches on alternating HIP streams (round 6).  A synthesis step is a chain: the prior transformers and the flow -- ~250 short,
    latency-bound launches that leave most of the chip idle -- then the HiFi-GAN generator, whose launches fill it.  Batches are independent (no op of the path
    mixes utterances), so batch i + 1 issued on a second stream runs its transformers in the gaps of batch i's generator: measured on one
    MI355X, B = 32 x T_mel = 1024: 72.0 -> 69.3 ms a batch; BASELINE configs[1] (B = 8, T_mel = 512): 9.85 -> 8.31 ms; configs[4]: 56.7 -> 54.2 ms
    (profiles/r06_stream_rotation_ab.txt); a third stream adds nothing.  Every launch of the library goes to torch's current stream and every workspace is
    allocated through torch's stream-aware allocator, so two steps in flight share only read-only state (parameters, packed weights).  NOT for the flow's
    FORWARD direction with log-det (its partial sums live in the conv handle: INTEGRATION.md 2) nor for training."""

    def __init__(self, n=2, timing=False):
        self.streams = [torch.cuda.Stream() for _ in range(max(1, int(n)))]
        self.count = 0
        self.timing = bool(timing)                # (events that can be timed against each other: the benchmark's per-batch statistics)
        cur = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(cur)                  # (inputs prepared on the caller's stream so far are visible)

    def run(self, fn):
        """fn() with the next stream of the rotation current; returns (fn's result, an event recorded behind it on that stream)"""
        st = self.streams[self.count % len(self.streams)]
        self.count += 1
        with torch.cuda.stream(st):
            out = fn()
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record(st)
        return out, ev

    def join(self):
        """the caller's current stream waits for everything issued through the rotation"""
        cur = torch.cuda.current_stream()
        for st in self.streams:
            cur.wait_stream(st)


class GraphedStep:
    """One synthesis step (VISinger.forward(infer=True)) of a FIXED shape captured into a HIP graph and replayed: what a serving
    loop with recurring batch shapes does.  Every launch of the step goes to torch's current stream through the C ABI, nothing
    allocates with hipMalloc or synchronises with the host after the warm-up, so the ~700 launches of a step replay as one graph
    (tests/test_model_gpu.py::test_synthesis_step_is_graph_capturable).  Inputs are copied into the graph's static buffers."""

    @staticmethod
    def fingerprint(model):
        """(data_ptr, in-place version) of every parameter: changes with optimizer steps, load_state_dict and copy_ (not with edits
        through `.data`, like the packed-weight caches: call hipconv.repack_weights AND drop the graphs after such an edit)"""
        return tuple((p.data_ptr(), p._version) for p in model.parameters())

    def __init__(self, model, batch, noise, mask_decoder):
        self.weights = self.fingerprint(model)       # a replay never re-folds / re-packs weights: the graph is only valid for these
        self.static = {k: v.clone() for k, v in batch.items()}
        self.noise = noise.clone()

        def run():
            b = self.static
            return model(b["text_tokens"], b["pitch_tokens"], b["dur_tokens"], b["mel2ph"], spk_id=b["spk_id"], infer=True,
                         noise=self.noise, mask_decoder=mask_decoder)["wav_out"]

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # warm-up off the capture: packs weights, sizes every workspace
            run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = run()

    def __call__(self, batch, noise):
        for k, v in batch.items():
            self.static[k].copy_(v)
        self.noise.copy_(noise)
        self.graph.replay()
        return self.out          # the graph's STATIC output buffer: the next replay overwrites it (clone to keep it)


@torch.no_grad()
def synthesize(model, items, hop_size, max_frames_per_batch=32768, noise_scale=1.0, generator=None, equal_tokens=False,
               graphs=None, streams=2):
    """Run VISinger.forward(infer=True) over length-bucketed batches.  Returns a list of float32 waveforms trimmed to
    each item's own length (frames * hop_size), in the input order.

    Against the item's one-at-a-time synthesis (the reference's test_step, tasks/visinger.py:244-263) on the same noise:
      * the prior, the flow and the attention are masked per item by the reference itself;
      * the HiFi-GAN generator -- which the reference runs unmasked, on one utterance -- is run with the frame mask at every stage
        (Generator.forward x_mask) whenever a batch holds items of different lengths, so nothing leaks from the padding into an
        item's last frames;
      * the reference's TextEncoder views its positional-embedding table by the PADDED token count (encoder.py:52-54, a `seq_len =
        hidden` mix-up restated literally): an item padded to a longer token sequence gets a different embedding than alone.
        `equal_tokens=True` lets only items of equal token count share a batch -- then every waveform EQUALS its one-at-a-time
        synthesis; the default batches by frames only and is exact for the items that define a batch's token length.

    graphs: a dict owned by the caller; when given, each batch shape (B, T_tokens, T_frames, ragged) is captured into a HIP graph
    on first use (GraphedStep) and replayed afterwards -- the launch chain of a small batch (B=1: ~700 dependent launches) then
    costs one graph launch instead of ~700 host-side launches.

    streams: consecutive batches go to alternating HIP streams (StreamRotation: the next batch's transformers run under this batch's generator); a batch's
    waveforms come back to the host when `streams` later batches have been issued.  The result is bit-identical to streams = 1 (same kernels, same inputs: the
    noise of every batch is drawn on the caller's stream, in batch order).  With `graphs` the batches replay on the caller's stream (one stream)."""
    device = next(model.parameters()).device
    lengths = [int((np.asarray(it["mel2ph"]) > 0).sum()) for it in items]
    out = [None] * len(items)
    keys = [len(it["text_tokens"]) for it in items] if equal_tokens else None
    rotation = StreamRotation(streams) if (graphs is None and streams and streams > 1) else None
    pending = []                                 # (item indices, device waveforms, event) of the batches in flight

    def collect(idx, wav_dev, ev):
        if ev is not None:
            ev.synchronize()
        wav = wav_dev.float().cpu().numpy()
        for b, i in enumerate(idx):
            out[i] = wav[b, :lengths[i] * hop_size].copy()

    for idx in bucket_by_length(lengths, max_frames_per_batch, keys=keys):
        batch = collate([items[i] for i in idx], device)
        B, T = batch["mel2ph"].shape
        noise = torch.randn((B, model.hidden_size, T), device=device, generator=generator) * noise_scale
        ragged = len({lengths[i] for i in idx}) > 1
        if graphs is not None:
            key = (B, batch["text_tokens"].shape[1], T, ragged)
            if key not in graphs or graphs[key].weights != GraphedStep.fingerprint(model):      # (re-captured after a weight update)
                graphs[key] = GraphedStep(model, batch, noise, ragged)
            collect(idx, graphs[key](batch, noise), None)
            continue

        def run(batch=batch, noise=noise, ragged=ragged):
            return model(batch["text_tokens"], batch["pitch_tokens"], batch["dur_tokens"], batch["mel2ph"],
                         spk_id=batch["spk_id"], infer=True, noise=noise, mask_decoder=ragged)["wav_out"]

        if rotation is None:
            collect(idx, run(), None)
            continue
        for st in rotation.streams:              # (this batch's inputs were made on the caller's stream)
            st.wait_stream(torch.cuda.current_stream())
        wav_dev, ev = rotation.run(run)
        for t in list(batch.values()) + [noise]:
            t.record_stream(rotation.streams[(rotation.count - 1) % len(rotation.streams)])      # (their memory is reused only behind that stream's work)
        pending.append((idx, wav_dev, ev))
        if len(pending) > len(rotation.streams):
            collect(*pending.pop(0))
    for job in pending:
        collect(*job)
    if rotation is not None:
        rotation.join()
    return out
