"""The OMP355_* environment knobs (A/B switches of the sweeps under tools/ and profiles/): read through these helpers so that a typo is an
error message, not a silently different engine.  Knobs that change an engine's LAYOUT (OMP355_KV_SPLIT) are part of the engine key
(model/omniparser.py::_key); the others change scheduling only.

  OMP355_ENC_CHUNK     images per encoder pass inside one engine call (0 = by engine, the default: 80 bf16 / fp32, 54 bf16x3 -- whole rounds of the
                       stage-2 chains, model/omniparser.py::enc_chunk_default; 1..4096 forces a size)
  OMP355_KV_SPLIT      bf16x3 engine: split-plane K / V^T slabs (1, default) or fp32 slabs (0)
  OMP355_KV_ROWS       bf16 engine: memory projection as one row-owner launch per tensor (1, default) or the two tiled GEMMs (0)
  OMP355_CROSS_SPLIT   override of the cross-attention key split (0 = automatic; a power of two <= 16)
  OMP355_ROWS_MIN      rows from which a decoder phase runs its Linear layers as row-owner chains (default 4096)
  OMP355_MID_MIN       rows from which the launch-per-Linear step of the bf16 engine runs its mid section as a chain (default 64; 1073741824 = never)
  OMP355_MLP_ROWS_MIN  tokens from which the blocks of Swin-B's stage 2 run as row-owner chains (default 32768)
  OMP355_PAIR          polygon || recognition phases on the chains as ONE interleaved schedule with serialised cross-attention launches
                       (omp_decoder_run_pair; 0 / 1, default 0: measured equal to slower, profiles/r06b_*, r06c_*)
  OMP355_SIDE_PRIO     the model's polygon / recognition side streams at high priority (0 / 1, default 0: they cost the pipelined lanes their hardware queues)
  OMP355_XCD_SPLIT     each decoder's many-row chains on its own four XCDs (0 / 1, default 0: chain HBM traffic 1.30x -> 1.14x algorithmic, but the
                       phase 98 -> 109 ms: the other decoder's cross-attention is left with four XCDs' share of the fabric; profiles/r06d_*)
  OMP355_X3_MIN        bf16x3 engine: rows from which a decoder phase runs its products as split-bf16 GEMMs (default 65; 161 puts the 160-row point phase
                       on the fp32 few-row kernels: measured slower, profiles/r06y_*)
  OMP355_DEC_PRIORITY  pipeline lanes: decoder streams at high priority (0 / 1, default 0)
Read in C (csrc/decoder.hip), parsed strictly there -- anything but a non-negative integer fails every omp_decoder_run:
  OMP355_SAMPLE_BLOCK_MAX_ROWS  rows up to which sampling runs a workgroup per row (default 1024)
  OMP355_FUSED_SA_MAX_ROWS      rows up to which the fused few-row self-attention kernel runs (default 63)
  OMP355_GRAPH_RUN              8 (default): runs of 8 sampling steps replay as ONE hipGraph; 1: one graph launch per step (A/B)"""
import os


def env_int(name, default, lo, hi, allowed=None):
    raw = os.environ.get(name)
    if raw is None or raw == '':
        return default
    try:
        v = int(raw)
    except ValueError:
        raise ValueError('%s=%r is not an integer' % (name, raw))
    if v < lo or v > hi or (allowed is not None and v not in allowed):
        raise ValueError('%s=%d is outside %s' % (name, v, sorted(allowed) if allowed is not None else '[%d, %d]' % (lo, hi)))
    return v


def env_flag(name, default):
    raw = os.environ.get(name)
    if raw is None or raw == '':
        return bool(default)
    if raw not in ('0', '1'):
        raise ValueError('%s=%r must be 0 or 1' % (name, raw))
    return raw == '1'
