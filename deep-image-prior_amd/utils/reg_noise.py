"""Input-noise regularisation of the closures on the device RNG of libdip_hip.so (no counterpart
file in the reference, which writes it inline):

    denoising.ipynb:198,208-209      noise = net_input.detach().clone()
                                     net_input = net_input_saved + (noise.normal_() * reg_noise_std)

    reg = RegNoise(net_input_saved, reg_noise_std, seed=0)
    def closure():
        net_input = reg()                 # net_input_saved + N(0,1) * reg_noise_std, ONE launch

Counter-based Philox4x32-10 + Box-Muller (dip_noise_axpy_dev); the stream position lives in device
memory and advances with every call, so the call is a static launch (hipGraph-replayable) and no
normal_() + mul + add temporaries are made.  The stream is NOT torch's device generator stream (the
reference's device stream cannot be reproduced on a CPU oracle either, SURVEY.md section 8c "RNG").
"""
import torch

import dip_native as N


class RegNoise:
    def __init__(self, net_input_saved, reg_noise_std, seed=0):
        if not net_input_saved.is_cuda:
            raise RuntimeError("dip-amd: RegNoise works on MI355X tensors only (no CPU fallback)")
        self.saved = net_input_saved.detach().contiguous().float()
        self.std = float(reg_noise_std)
        self.seed = int(seed)
        self.offset = torch.zeros(1, dtype=torch.int64, device=self.saved.device)
        self.out = torch.empty_like(self.saved)

    def __call__(self):
        """Returns net_input_saved + N(0,1)*std in a buffer owned by this object (overwritten by the
        next call, like the reference's `noise` tensor)."""
        if self.std <= 0:
            return self.saved
        dev = self.saved.device
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            N.check(N.lib().dip_noise_axpy_dev(self.saved.data_ptr(), self.out.data_ptr(), self.saved.numel(),
                                               self.std, self.seed, self.offset.data_ptr(), st), "noise_axpy_dev")
        return self.out
