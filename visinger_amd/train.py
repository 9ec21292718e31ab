"""GAN training step of VISinger on the MI355X-native modules, restated from the reference's
``tasks/visinger.py:53-170,201-227`` (losses, optimizers, schedulers), ``tasks/base.py:227-238`` (masked mel L1) and
``utils/commons/trainer.py:306-384`` (two optimizer passes per batch with the other optimizer's parameters frozen,
gradient clipping over ALL task parameters, ``optimizer.zero_grad()`` right after every ``optimizer.step()``, and
``tasks/visinger.py:221-227`` (both learning rates set in closed form after every optimizer step: ``base * gamma ** epoch`` with
the config's ``endless_ds: false`` (config/models/visinger.yaml:106), ``base * gamma ** (global_step // accumulate)`` otherwise).

Forward: HIP conv engine (visinger_amd.autograd); backward: PyTorch-ROCm autograd; data parallel: stock
``DistributedDataParallel(find_unused_parameters=True)`` over RCCL, one process per GPU -- the gradient all-reduce (<= 430 MB
fp32 per backward) is the only collective of the training path.

Deviations from the reference, all forced by its latent bugs: `uv` is passed to the model (the reference
forgets it), ``lambda_uv`` / ``lambda_f0`` default to ``lambda_pitch`` (they are defined in no YAML).  The mel loss uses the
``torch.stft`` restatement of torchaudio's transforms (visinger_amd/audio.py: parity unpinned).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import audio
from .models.visinger import MultiPeriodDiscriminator, VISinger
from .modules.commons.utils import slice_segments

TRAIN_HPARAMS = dict(lr=2e-4, optimizer_adam_beta1=0.8, optimizer_adam_beta2=0.99, eps=1e-9, weight_decay=0.001,
                     discriminator_optimizer_params=dict(eps=1e-9, weight_decay=0.0), scheduler_gamma=0.999875,
                     clip_grad_norm=1.0, lambda_kl=1.0, kl_min=0.0, kl_start_steps=1, lambda_mel=45.0, lambda_ctc=45.0,
                     lambda_mel_adv=1.0, lambda_fm=2.0, lambda_pitch=10.0, disc_start_steps=0, disc_interval=1,
                     sample_rate=24000, fft_size=2048, win_size=1200, hop_size=300, fmin=20.0, fmax=12000.0,
                     num_mel_bins=128, use_spectral_norm=False, endless_ds=False, accumulate_grad_batches=1)


def masked_l1(pred, target):
    """tasks/base.py:232-238: L1 weighted by the frames whose target is not all-zero"""
    w = target.abs().sum(-1, keepdim=True).ne(0).float().expand_as(target)
    return (F.l1_loss(pred, target, reduction="none") * w).sum() / w.sum()


def discriminator_loss(tgt_outputs, gen_outputs):
    """tasks/visinger.py:147-153 (LSGAN)"""
    return sum(torch.mean((1 - t.float()) ** 2) + torch.mean(g.float() ** 2) for t, g in zip(tgt_outputs, gen_outputs))


def generator_loss(gen_outputs):
    """tasks/visinger.py:155-160"""
    return sum(torch.mean((1 - g.float()) ** 2) for g in gen_outputs)


def feature_matching_loss(fmap_tgt, fmap_gen):
    """tasks/visinger.py:162-169"""
    from .autograd import l1_mean            # (one launch each way per term where sub / abs / mean and their backward ran eight)
    return sum(l1_mean(g.float(), t.float()) for ft, fg in zip(fmap_tgt, fmap_gen) for t, g in zip(ft, fg))


def pitch_losses(f0_pred, f0, uv, mel2ph, lambda_uv, lambda_f0):
    """tasks/visinger.py:127-139: voiced/unvoiced BCE over the valid frames, f0 L1 over the valid VOICED frames -> (uv loss, f0 loss).
    (The reference divides by the voiced-frame count unguarded; a batch without any voiced frame gives 0 here instead of NaN.)"""
    nonpad = (mel2ph != 0).float()
    uv_loss = (F.binary_cross_entropy_with_logits(f0_pred[:, :, 1], uv, reduction="none") * nonpad).sum() / nonpad.sum() * lambda_uv
    voiced = nonpad * (uv == 0).float()
    f0_loss = (F.l1_loss(f0_pred[:, :, 0], f0, reduction="none") * voiced).sum() / voiced.sum().clamp(min=1) * lambda_f0
    return uv_loss, f0_loss


def ctc_loss(ph_pred, text_tokens, mel_lengths, text_lengths, lambda_ctc):
    """tasks/visinger.py:140-145: CTC of the frame-level phoneme log-probabilities [B, dict, T] against the token sequence"""
    return F.ctc_loss(ph_pred.float().permute(2, 0, 1), text_tokens, mel_lengths, text_lengths, zero_infinity=True) * lambda_ctc


def kl_loss(kl, global_step, kl_start_steps, kl_min, lambda_kl):
    """tasks/visinger.py:103-107: clamp, linear warm-up over kl_start_steps, weight"""
    return min(global_step / kl_start_steps, 1) * torch.clamp(kl, min=kl_min) * lambda_kl


class VISingerTrainer(nn.Module):
    """Owns the two networks (children named ``model`` and ``mel_disc`` like the reference task, so the checkpoint layout
    matches), the two AdamW optimizers and their ExponentialLR schedulers."""

    def __init__(self, ph_dict_size, pitch_size, dur_size, hparams, train_hparams=None):
        super().__init__()
        self.hp = dict(TRAIN_HPARAMS, **(train_hparams or {}))
        self.hop = int(torch.tensor(hparams["upsample_rates"]).prod())
        self.hp["hop_size"] = self.hop
        self.segment_size = hparams["segment_size"]
        self.model = VISinger(ph_dict_size, pitch_size, dur_size, hparams)
        self.mel_disc = MultiPeriodDiscriminator(self.hp["use_spectral_norm"])
        self.global_step = 0
        self.current_epoch = 0        # advanced by on_epoch_end(): the learning rate decays once per EPOCH (endless_ds: false)
        self._cached = None

    def configure(self):
        h = self.hp
        # torch's FUSED AdamW where every parameter lives on the GPU (the same update rule, step counters on the device): the default (foreach)
        # implementation keeps one CPU step counter per parameter -- ~860 `.item()` / `add_` host round trips and ~470 small launches per optimizer
        # step (tools/train_launch_count.py, round 4).  VS_NO_FUSED_ADAMW=1: the A/B switch.
        from ._lib import switch
        fused = (not switch("VS_NO_FUSED_ADAMW")) and all(p.is_cuda for p in self.parameters())
        kw = dict(fused=True) if fused else {}
        self.opt_gen = torch.optim.AdamW(self.model.parameters(), lr=h["lr"], betas=(h["optimizer_adam_beta1"], h["optimizer_adam_beta2"]),
                                         weight_decay=h["weight_decay"], eps=h["eps"], **kw)
        self.opt_disc = torch.optim.AdamW(self.mel_disc.parameters(), lr=h["lr"],
                                          betas=(h["optimizer_adam_beta1"], h["optimizer_adam_beta2"]),
                                          **h["discriminator_optimizer_params"], **kw)
        self.sched = [torch.optim.lr_scheduler.ExponentialLR(o, gamma=h["scheduler_gamma"]) for o in (self.opt_gen, self.opt_disc)]
        return self

    def lr_exponent(self):
        """tasks/visinger.py:221-227: the argument the reference passes to ``scheduler.step(...)``"""
        if self.hp["endless_ds"]:
            return self.global_step // self.hp["accumulate_grad_batches"]
        return self.current_epoch

    def on_after_optimization(self):
        """``ExponentialLR.step(n)`` with an explicit n is the closed form lr = base_lr * gamma ** n (torch's
        ``_get_closed_form_lr``); the reference calls it for BOTH schedulers after every optimizer step."""
        n = self.lr_exponent()
        for sch in self.sched:
            sch.last_epoch = n
            for group, base in zip(sch.optimizer.param_groups, sch.base_lrs):
                group["lr"] = base * sch.gamma ** n
            sch._last_lr = [g["lr"] for g in sch.optimizer.param_groups]

    def on_epoch_end(self):
        """trainer.py:300-301: one pass over the training set is over"""
        self.current_epoch += 1

    def mel(self, wav):
        h = self.hp
        return audio.mel_spectrogram(wav, h["sample_rate"], h["fft_size"], h["win_size"], h["hop_size"], h["num_mel_bins"],
                                     h["fmin"], h["fmax"])            # [B, T, n_mels]

    # -- the two passes of tasks/visinger.py:53-89 ----------------------------------------------------------------
    def generator_pass(self, batch):
        h = self.hp
        # (noise_q / u_slice: optional injected RNG draws -- posterior noise [B, H, T] and segment-start uniforms [B] -- so that a sharded
        #  run and a single-process run of the same global batch see the same random numbers: tests/test_ddp_two_ranks_gpu.py)
        out = self.model(batch["text_tokens"], batch["note_pitch"], batch["note_dur"], batch["mel2ph"], spk_id=batch.get("spk_ids"),
                         f0=batch.get("f0"), uv=batch.get("uv"), mel=batch["mels"], infer=False, noise_q=batch.get("noise_q"),
                         u_slice=batch.get("u_slice"))
        losses = {"kl": kl_loss(out["kl"], self.global_step, h["kl_start_steps"], h["kl_min"], h["lambda_kl"])}
        tgt_mel = self.mel(batch["wavs"])                                                   # [B, T, M]
        tgt_slice = slice_segments(tgt_mel.transpose(1, 2).contiguous(), out["ids_slice"], self.segment_size).transpose(1, 2)
        losses["mel_l1"] = masked_l1(self.mel(out["wav_out"]), tgt_slice) * h["lambda_mel"]
        if "f0_pred" in out and batch.get("f0") is not None:
            losses["uv"], losses["f0"] = pitch_losses(out["f0_pred"], batch["f0"], batch["uv"], batch["mel2ph"], h["lambda_pitch"], h["lambda_pitch"])
        if "ph_pred" in out:
            losses["ctc"] = ctc_loss(out["ph_pred"], batch["text_tokens"], batch["mel_lengths"], batch["text_lengths"], h["lambda_ctc"])
        self._cached = {k: v.detach() for k, v in out.items() if isinstance(v, torch.Tensor)}
        if self.global_step >= h["disc_start_steps"] and h["lambda_mel_adv"] > 0:
            real = slice_segments(batch["wavs"].unsqueeze(1), out["ids_slice"] * self.hop, self.segment_size * self.hop)
            _, d_gen, fmap_tgt, fmap_gen = self.mel_disc(real, out["wav_out"].unsqueeze(1))
            losses["generator"] = generator_loss(d_gen) * h["lambda_mel_adv"]
            losses["feature_match"] = feature_matching_loss(fmap_tgt, fmap_gen) * h["lambda_fm"]
        return sum(losses.values()), losses

    def discriminator_pass(self, batch):
        out = self._cached
        real = slice_segments(batch["wavs"].unsqueeze(1), out["ids_slice"] * self.hop, self.segment_size * self.hop)
        d_tgt, d_gen, _, _ = self.mel_disc(real, out["wav_out"].unsqueeze(1))
        loss = discriminator_loss(d_tgt, d_gen)
        return loss, {"discriminator": loss}

    def forward(self, batch, optimizer_idx):
        """DDP entry point (the reference routes DDP.forward to training_step, ddp_utils.py:75-80)"""
        banks = self._weight_banks(optimizer_idx)
        try:
            for bank in banks:
                bank.refresh()
            return self.generator_pass(batch) if optimizer_idx == 0 else self.discriminator_pass(batch)
        finally:
            for bank in banks:
                bank.release()

    def _weight_banks(self, optimizer_idx):
        """the networks whose weights the pass derives: folded and packed in a handful of launches before the forward (weight_bank.py)"""
        from ._lib import switch
        if switch("VS_NO_WEIGHT_BANK") or not (self.training and torch.is_grad_enabled()) or not self._param_lists()[0][0].is_cuda:
            return ()
        if self.__dict__.get("_banks") is None:
            from .weight_bank import WeightBank
            self.__dict__["_banks"] = (WeightBank(self.model), WeightBank(self.mel_disc))
        bg, bd = self.__dict__["_banks"]
        disc_on = self.global_step >= self.hp["disc_start_steps"] and self.hp["lambda_mel_adv"] > 0
        return ((bg, bd) if disc_on else (bg,)) if optimizer_idx == 0 else (bd,)

    def __getstate__(self):                  # (the per-parameter-set caches hold device buffers and table pointers: process-local, rebuilt on demand)
        state = self.__dict__.copy()
        state.pop("_plists", None)
        state.pop("_banks", None)
        state.pop("_structure_seen", None)
        return state

    def train(self, mode=True):
        """(also drops the per-parameter-set caches -- parameter lists, weight banks: structural surgery such as remove_weight_norm happens between mode switches)"""
        self.__dict__.pop("_plists", None)
        self.__dict__.pop("_banks", None)
        return super().train(mode)

    def _param_lists(self):
        """(generator parameters, discriminator parameters) as lists, built once per parameter set: a step walks them seven times (requires_grad of both
        networks per pass, the clip norm, the restore) and `Module.parameters()` re-walks ~700 modules each time -- 5 ms of host time a step on a path that is
        as much host- as device-bound"""
        c = self.__dict__.get("_plists")
        n = sum(1 for _ in self._modules)      # (cheap guard against a swapped sub-network)
        if c is None or c[2] != (id(self.model), id(self.mel_disc), n):
            c = (list(self.model.parameters()), list(self.mel_disc.parameters()), (id(self.model), id(self.mel_disc), n))
            self.__dict__["_plists"] = c
        return c[0], c[1]

    def backward_pass(self, batch, opt_idx, runner=None):
        """forward + backward of one optimizer's pass with the other network frozen (trainer.py:312-375): afterwards the gradients of
        `own` are in place -- under DDP already all-reduced (averaged over the ranks)."""
        runner = runner or self
        pg, pd = self._param_lists()
        own, other = (pg, pd) if opt_idx == 0 else (pd, pg)
        for p in other:
            p.requires_grad_(False)
        for p in own:
            p.requires_grad_(True)
        loss, parts = runner(batch, opt_idx)
        loss.backward()                          # under DDP: the bucketed gradient all-reduce over RCCL happens here
        return parts

    # Python's cyclic collector and the launch queue: a step allocates ~10^5 Python objects (autograd nodes, ctypes argument blocks), and a
    # generation-2 pass in the middle of one stalls the host while the device drains -- 0.3-3 ms a step in A/B runs.  The trainer therefore keeps the collector
    # OUT of a step and runs it BETWEEN steps every `gc_every` steps (what large training loops do with gc.freeze / manual collection); the benchmark times this
    # same method, so what it reports is what a user gets.  gc_every = 0 leaves the collector alone.
    gc_every = 50

    def training_step(self, batch, runner=None):
        """One iteration = generator pass + discriminator pass (trainer.py:306-384).  `runner` is the (optionally
        DDP-wrapped) module to call; gradients are clipped over ALL parameters of the task, as the reference does."""
        import gc
        manage_gc = bool(self.gc_every) and gc.isenabled()
        if manage_gc:
            if self.global_step % self.gc_every == 0:
                gc.collect()
            gc.disable()
        try:
            return self._training_step(batch, runner)
        finally:
            if manage_gc:
                gc.enable()

    def _structure(self):
        """fingerprint of the task's parameter set: one walk over the modules a step (~0.5 ms) where _param_lists() would otherwise trust its cache blindly"""
        n = first = last = 0
        for m in self.modules():
            for p in m._parameters.values():
                if p is not None:
                    n += 1
                    last = id(p)
                    first = first or last
        return n, first, last

    def _training_step(self, batch, runner=None):
        from .autograd import bump_weight_epoch
        bump_weight_epoch()      # edits made through p.data since the last step (EMA swaps) must not meet a stale packed weight
        # (parameter-level surgery without a train() / eval() toggle -- remove_weight_norm, added or swapped parameters -- must not leave the cached
        #  parameter lists and weight banks acting on orphaned Parameters)
        fp = self._structure()
        if self.__dict__.get("_structure_seen") != fp:
            self.__dict__.pop("_plists", None)
            self.__dict__.pop("_banks", None)
            self.__dict__["_structure_seen"] = fp
        logs = {}
        for opt_idx, opt in enumerate((self.opt_gen, self.opt_disc)):
            parts = self.backward_pass(batch, opt_idx, runner)
            if self.hp["clip_grad_norm"] > 0:    # (the other network holds no gradients: zeroed right after ITS step, as below)
                torch.nn.utils.clip_grad_norm_(self._param_lists()[0] + self._param_lists()[1], self.hp["clip_grad_norm"])
            opt.step()
            opt.zero_grad(set_to_none=True)      # trainer.py:373-374: no stale gradients in the next pass's clip norm
            self.on_after_optimization()
            logs.update({k: v.detach() for k, v in parts.items()})
        for plist in self._param_lists():
            for p in plist:
                p.requires_grad_(True)
        self.global_step += 1
        # the loss values go to the host ONCE, at the end of the step: reading the generator pass's values before launching the
        # discriminator pass drained the queue in the middle of every step (tools/train_phases.py: 15 ms of host wait, after which the
        # discriminator pass started on an idle GPU)
        return {k: float(v) for k, v in logs.items()}


def synthetic_train_batch(B, T, Tph, hop, ph_dict, n_bins, seed, device):
    """BASELINE config-3 synthetic batch: tokens uniform, mel2ph = repeat_interleave, linear spectrogram
    |N(0,1)|^2, waveform U(-0.5, 0.5), log-f0 / uv random."""
    g = torch.Generator().manual_seed(seed)
    b = dict(text_tokens=torch.randint(4, ph_dict, (B, Tph), generator=g), note_pitch=torch.randint(1, 117, (B, Tph), generator=g),
             note_dur=torch.randint(4, 131, (B, Tph), generator=g),
             mel2ph=torch.repeat_interleave(torch.arange(1, Tph + 1), T // Tph)[None].repeat(B, 1),
             mels=torch.randn(B, T, n_bins, generator=g).abs() ** 2, wavs=torch.rand(B, T * hop, generator=g) - 0.5,
             f0=torch.rand(B, T, generator=g) * 2 + 4, uv=(torch.rand(B, T, generator=g) < 0.2).float(),
             spk_ids=torch.zeros(B, dtype=torch.long), mel_lengths=torch.full((B,), T, dtype=torch.long),
             text_lengths=torch.full((B,), Tph, dtype=torch.long))
    return {k: v.to(device) for k, v in b.items()}
