"""MI355X-native TCAR training path (hand-written HIP kernels behind a C-ABI, host mirror of the
reference's Seq2SeqAttNN / Sampler interface).  See DESIGN.md."""
__version__ = "0.1.0"
