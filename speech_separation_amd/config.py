"""Model hyper-parameters of the MossFormer2_SS separation network.

The attribute names read off ``args`` are the ones the reference's model
factory consumes (reference: models/mossformer2/mossformer2.py:742-753,
config/inference/MossFormer2_SS_8K.yaml:16-20).  Everything else is fixed by
the reference's constructors (group size 256, query/key dim 128, expansion 4,
TCN depth 8, depthwise kernel 17; reference:
models/mossformer2/mossformer2_block.py:612-643, conv_module.py:282-296).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class MossFormer2Config:
    dim: int = 512            # encoder_embedding_dim == mossformer_sequence_dim
    num_layers: int = 24      # num_mossformer_layer
    enc_kernel: int = 16      # encoder_kernel_size (stride = kernel // 2)
    num_spks: int = 2
    group_size: int = 256
    qk_dim: int = 128
    rot_dim: int = 32         # min(32, qk_dim)
    tcn_depth: int = 8
    dw_kernel: int = 17
    # The block interleaved with the FLASH layers: "tcn" = GatedTCNBlock (what the reference revision builds,
    # mossformer2_block.py:635-643), "fsmn" = Gated_FSMN_Block_Dilated (:430-463; the upstream architecture's block,
    # defined in the reference but not instantiated by it), with fsmn_inner inner channels.
    recurrent: str = "tcn"
    fsmn_inner: int = 256

    @property
    def hidden(self) -> int:      # to_hidden output = v | u
        return 4 * self.dim

    @property
    def vdim(self) -> int:        # width of v (and of u)
        return 2 * self.dim

    @property
    def stride(self) -> int:
        return self.enc_kernel // 2

    def frames(self, T: int) -> int:
        return (T - self.enc_kernel) // self.stride + 1

    def padded_frames(self, T: int) -> int:
        S = self.frames(T)
        g = self.group_size
        return (S + g - 1) // g * g

    def flops(self, T: int) -> float:
        """Algorithmic FLOPs of one utterance of T samples (GEMM-shaped work only)."""
        S, Sp = self.frames(T), self.padded_frames(T)
        d, h, q, k17 = self.dim, self.hidden, self.qk_dim, self.dw_kernel
        v2 = 2 * self.vdim
        flash_s = d * h + h * k17 + d * q + q * k17 + q * v2 + q * v2 + self.vdim * d + d * k17
        flash_sp = q * self.group_size + self.group_size * v2
        tcn = 3 * d * d + self.tcn_depth * (3 * d + 2 * d * d)
        if self.recurrent == "fsmn":       # conv1, to_u | to_v (+ ConvModule), linear, project, the two 39-tap convs, conv2
            i = self.fsmn_inner
            tcn = d * i + 2 * i * i + 2 * i * k17 + 2 * i * i + 39 * i + 2 * 39 * i + i * d
        head = d * d + d * d * self.num_spks + 2 * d * d * self.num_spks + d * d * self.num_spks \
            + self.enc_kernel * d + self.enc_kernel * d * self.num_spks
        macs = self.num_layers * ((flash_s + tcn) * S + flash_sp * Sp) + head * S
        return 2.0 * macs

    @staticmethod
    def from_args(args) -> "MossFormer2Config":
        dim = int(args.encoder_embedding_dim)
        if int(args.mossformer_sequence_dim) != dim:
            raise ValueError("encoder_embedding_dim must equal mossformer_sequence_dim "
                             "(the reference multiplies the mask with the encoder output)")
        return MossFormer2Config(dim=dim,
                                 num_layers=int(args.num_mossformer_layer),
                                 enc_kernel=int(args.encoder_kernel_size),
                                 num_spks=int(args.num_spks),
                                 recurrent=str(getattr(args, "msf2_recurrent", "tcn")),
                                 fsmn_inner=int(getattr(args, "msf2_fsmn_inner", 256)))


FULL_8K = MossFormer2Config()
# small configurations used by the parity fixtures
TINY = MossFormer2Config(dim=64, num_layers=2)
SMALL = MossFormer2Config(dim=128, num_layers=2)
TINY_FSMN = MossFormer2Config(dim=64, num_layers=2, recurrent="fsmn", fsmn_inner=64)
SMALL_FSMN = MossFormer2Config(dim=128, num_layers=2, recurrent="fsmn", fsmn_inner=64)
