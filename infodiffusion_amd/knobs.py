"""Every environment switch of the package, in one table (read once, at import).  They exist for same-box A/B measurements
(tools/ab_env.sh) and for the tests that pin both sides of a switch; the defaults are the measured-faster settings
(INTEGRATION.md section 5 has the same table with the measurement behind each default).  The library reads seven more through
its own table (csrc/idf_capi.hip: IDF_CONV_RS, IDF_CONV_RS_SYNC, IDF_CONV_PS, IDF_CONV_DLDS_MIN, IDF_WGRAD_KR3, IDF_WGRAD_TPB3, IDF_WGRAD_RING)."""
import os

TABLE = {
    # name: (default, what it switches)
    'IDF_LIB': ('', 'path of the HIP library to load instead of the in-tree libinfodiff_hip.so (variant builds: tools/build_variant.sh)'),
    'IDF_CONV_RS': ('1', '64x64 / 32x32 ResBlock convs in the register-weights / row-reuse form (0: halo / direct-to-LDS kernels; the library '
                         'reads the same name: 1 = two 256-thread workgroups per CU, 2 = one of 512)'),
    'IDF_CONV_RS_FWD': ('0', '... for every covered forward conv in training too (default: channel-changing convs, and every shape in inference)'),
    'IDF_CONV_WR': ('1', '16x16 / 8x8 per-op convs with fragment-major weights in registers (idf_conv_wr_*)'),
    'IDF_RB_SMALL': ('1', 'image-resident 8x8 ResBlock: forward one launch, backward one launch (idf_resblock_small_*)'),
    'IDF_RB_SMALL_MAXB': ('256', 'largest batch the image-resident block runs at'),
    'IDF_GN_FUSE': ('1', 'GroupNorm / FiLM / SiLU / dropout as the consuming conv\'s prologue (0: GroupNorm kernel + plain conv)'),
    'IDF_BWD_CHAIN': ('1', 'big-map backward: du epilogue in the data-gradient conv + streaming apply (0: one-launch GroupNorm backward)'),
    'IDF_BWD_LAZY': ('0', 'big-map backward: the NEXT data-gradient conv forms dy from (du, partials) in its prologue (built, tested, slower)'),
    'IDF_DGRAD_GN': ('1', 'small-map backward: GroupNorm backward as the data-gradient conv\'s epilogue'),
    'IDF_SC_FUSE': ('1', 'a block\'s 1x1 shortcut (and its data gradient) rides in its first conv\'s launches on the small maps'),
    'IDF_WGRAD_BATCH': ('1', 'weight gradients deferred to the end of the backward pass and launched as table-driven batches'),
    'IDF_DETERMINISTIC': ('1', 'bit-reproducible training steps (the reference\'s seed_everything sets cudnn.deterministic, utils.py:64-71): weight '
                               'gradients through per-split slabs + one ordered reduce launch, GroupNorm parameter gradients through per-image rows + one '
                               'ordered reduce launch at the end of the pass (0: fp32 atomics for both; 8.42 vs 8.44-8.47 ms per CelebA step)'),
    'IDF_ATTN_FOLD': ('1', 'AttnBlock: proj conv folded into V (Wv\' = Wp Wv)'),
    'IDF_ATTN_BLOCK_MINB': ('256', 'batch from which the 16x16 AttnBlock runs as ONE launch (idf_attnblock_fwd)'),
    'IDF_UPCONV': ('1', 'UpSample conv / its gradients as four 2x2 sub-pixel convs with summed weights'),
    'IDF_TEMB_FUSED': ('1', 'TimeEmbedding + fc_a + all FiLM projections behind one entry (idf_temb_film_*)'),
    'IDF_TRAJ_CACHE': ('1', 'samplers: the conditioning path (TimeEmbedding, fc_a, FiLM projections) once per trajectory -- a table over the timesteps + the latent\'s projections -- instead of once per step'),
    'IDF_SAMPLER_GRAPH': ('1', 'samplers replay ONE captured denoising step for the inner steps'),
    'IDF_SAMPLER_GRAPH_MAXPIX': (str(256 * 64 * 64), 'largest batch x H x W whose step is captured'),
    'IDF_SAMPLER_GRAPH_STRICT': ('0', 'a failed step capture raises instead of falling back to eager stepping'),
    'IDF_FORCE_SYNC': ('0', 'bench.py: run the data-parallel exchange path on ONE GPU (RCCL world size 1)'),
    'IDF_CPU_THREADS': ('16', 'bench.py: threads of the CPU baseline'),
}

_VAL = {k: os.environ.get(k, d) for k, (d, _) in TABLE.items()}


def raw(name):
    return _VAL[name]


def flag(name):
    return _VAL[name] not in ('0', '')


def num(name):
    return int(_VAL[name])
