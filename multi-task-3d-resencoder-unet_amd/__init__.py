"""MI355X-native engine for the multi-task 3-D residual-encoder U-Net hot path.

Drop-in surface (same names / argument meaning as the reference):
    mt3d_amd.builders.build_network_from_config.NetworkFromConfig(mgr)
    mt3d_amd.configuration.config_manager.ConfigManager(config_file)
    mt3d_amd.train.BaseTrainer(config_file, verbose=True, debug_dataloader=False)

The forward/backward of the network runs entirely on hand-written gfx950 kernels behind the C ABI
of `include/rxunet.h` (`csrc/librxunet.so`).  There is no PyTorch / CPU fallback: without the
library or without a gfx950 device, the compute entry points raise.
"""
__version__ = "0.1.0"
