"""Oracle restatement of mod_extraction/losses.py (TEST INFRASTRUCTURE ONLY), torch on the CPU.

l1 / mse are ``torch.nn`` modules exactly as in losses.py:143-150; ESR / DC follow losses.py:14-67;
fdl1 / sdl1 follow losses.py:70-102 (central differences ``(x[i+2] - x[i]) / 2``).  ``mrstft`` restates
auraloss==0.4.0 ``MultiResolutionSTFTLoss`` defaults (third-party, absent: PARITY UNPINNED).
"""
import torch
from torch import Tensor as T, nn


def central_diff(x: T) -> T:
    assert x.size(-1) > 2
    return (x[..., 2:] - x[..., :-2]) / 2.0


class FirstDerivativeL1Loss(nn.Module):
    def forward(self, input: T, target: T) -> T:
        return nn.functional.l1_loss(central_diff(input), central_diff(target))


class SecondDerivativeL1Loss(nn.Module):
    def forward(self, input: T, target: T) -> T:
        return nn.functional.l1_loss(central_diff(central_diff(input)), central_diff(central_diff(target)))


class ESRLoss(nn.Module):
    def __init__(self, eps: float = 1e-8) -> None:
        super().__init__()
        self.eps = eps

    def forward(self, input: T, target: T) -> T:
        return (((target - input) ** 2).sum(dim=-1) / ((target ** 2).sum(dim=-1) + self.eps)).mean()


class DCLoss(nn.Module):
    def __init__(self, eps: float = 1e-8) -> None:
        super().__init__()
        self.eps = eps

    def forward(self, input: T, target: T) -> T:
        return (((target - input).mean(dim=-1) ** 2) / ((target ** 2).mean(dim=-1) + self.eps)).mean()


class MultiResolutionSTFTLoss(nn.Module):
    """auraloss 0.4.0 defaults: fft (1024, 2048, 512), hop (120, 240, 50), win (600, 1200, 240), hann,
    w_sc = w_log_mag = 1, w_lin_mag = w_phs = 0, eps 1e-8, mean over resolutions.
    mag = sqrt(clamp(re^2 + im^2, min=eps)); sc = ||Y - X||_F / ||Y||_F; log-mag L1."""

    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                 eps: float = 1e-8) -> None:
        super().__init__()
        self.cfg, self.eps = list(zip(fft_sizes, hop_sizes, win_lengths)), eps

    def _mag(self, x: T, n_fft: int, hop: int, win: int) -> T:
        s = torch.stft(x.reshape(-1, x.size(-1)), n_fft, hop, win, torch.hann_window(win), return_complex=True)
        return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=self.eps))

    def forward(self, input: T, target: T) -> T:
        total = 0.0
        for n_fft, hop, win in self.cfg:
            xm, ym = self._mag(input, n_fft, hop, win), self._mag(target, n_fft, hop, win)
            sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
            lm = nn.functional.l1_loss(torch.log(xm), torch.log(ym))
            total = total + sc + lm
        return total / len(self.cfg)


class LogMelLoss(nn.Module):
    """losses.py:105-130: L1 between log(clip(MelSpectrogram(x), eps)) of input and target (torchaudio MelSpectrogram
    sample_rate 44100, n_fft 1024, hop 256, 256 HTK mels, power 2, centre -- the front end of models.py restated in
    oracle/models.py:MelFrontEnd.  That front end is torchaudio 0.13.1 arithmetic restated from its published
    definition: PARITY UNPINNED -- the CNN fixtures are generated with a name-only torchaudio stub and start AFTER the
    front end, so nothing reference-held checks the filter bank; only its torch.stft part is pinned)."""

    def __init__(self, sr: float = 44100, n_fft: int = 1024, hop_len: int = 256, n_mels: int = 256,
                 eps: float = 1e-7) -> None:
        super().__init__()
        from .models import MelFrontEnd
        self.eps = eps
        self.spectrogram = MelFrontEnd(int(sr), n_fft, hop_len, n_mels)

    def forward(self, input: T, target: T) -> T:
        a = torch.log(torch.clip(self.spectrogram(input), min=self.eps))
        b = torch.log(torch.clip(self.spectrogram(target), min=self.eps))
        return nn.functional.l1_loss(a, b)


def get_loss_func_by_name(name: str) -> nn.Module:
    table = {"l1": nn.L1Loss, "fdl1": FirstDerivativeL1Loss, "sdl1": SecondDerivativeL1Loss, "mse": nn.MSELoss,
             "esr": ESRLoss, "dc": DCLoss, "mrstft": MultiResolutionSTFTLoss, "log_mel_l1": LogMelLoss}
    if name not in table:
        raise KeyError(name)
    return table[name]()
