"""MI355X (gfx950) native hot path of the music-mixing-style-transfer inference pipeline.

    networks            FXencoder / TCNModel with the reference's module API, running on libmst_hip.so
    mixing_manipulator  FX-manipulator processors (EQ biquad cascade, compressor, imager, gain) on device
    inference           segment bookkeeping + style-transfer orchestration, multi-GPU segment sharding
    utils               deterministic synthetic weights / audio for tests and benchmarks
"""
__version__ = "0.1.0"
