"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Imports the *real* reference (`/root/reference`, read-only, exists only in the build
container, never on the GPU box) so that the oracle restatement in `oracle/resenc_oracle.py`
can be pinned against it and golden vectors can be generated (`oracle/make_golden.py`).

The reference's `builders/resblocks.py:9-11` imports the third-party package
`dynamic_network_architectures` (not installed, not vendored).  Three of the five imported
names have verbatim vendored twins inside the reference itself (`builders/utils.py:268-285`,
`:128-182`, `builders/simple_conv_blocks.py:13-79`); this shim aliases those.  `SqueezeExcite`
and `DropPath` have no twin -> constructors raise (parity unpinned for those two options).
"""
import contextlib
import io
import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("RX_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "builders", "build_network_from_config.py"))


def _install_dna_alias():
    if "dynamic_network_architectures.building_blocks.regularization" in sys.modules:
        return
    import builders.utils as ref_utils  # noqa: E402  (reference module)
    import builders.simple_conv_blocks as ref_scb  # noqa: E402

    names = ["dynamic_network_architectures",
             "dynamic_network_architectures.building_blocks",
             "dynamic_network_architectures.building_blocks.helper",
             "dynamic_network_architectures.building_blocks.simple_conv_blocks",
             "dynamic_network_architectures.building_blocks.regularization"]
    mods = {n: types.ModuleType(n) for n in names}
    mods[names[2]].maybe_convert_scalar_to_list = ref_utils.maybe_convert_scalar_to_list
    mods[names[2]].get_matching_pool_op = ref_utils.get_matching_pool_op
    mods[names[3]].ConvDropoutNormReLU = ref_scb.ConvDropoutNormReLU

    class _Unpinned:
        def __init__(self, *a, **k):
            raise NotImplementedError(
                "SqueezeExcite / DropPath live only in the un-vendored third-party package "
                "dynamic_network_architectures: parity unpinned")

    mods[names[4]].SqueezeExcite = _Unpinned
    mods[names[4]].DropPath = _Unpinned
    sys.modules.update(mods)


def import_reference():
    """Returns the reference's `NetworkFromConfig` class and its `losses` module."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True  # reference tree is read-only
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", SyntaxWarning)  # `is` with a literal, encoder.py:74-78
        _install_dna_alias()
        from builders.build_network_from_config import NetworkFromConfig  # noqa: E402
        import training.losses.losses as ref_losses  # noqa: E402
    return NetworkFromConfig, ref_losses


def make_mgr(patch_size, tasks, in_channels=1, batch_size=2, autoconfigure=True, model_config=None):
    """Plain attribute bag with exactly what build_network_from_config.py:21-32 reads."""
    return types.SimpleNamespace(
        tasks=tasks, train_patch_size=tuple(patch_size), train_batch_size=batch_size,
        in_channels=in_channels, vram_max=16.0, autoconfigure=autoconfigure,
        model_config=dict(model_config or {}))


def build_reference_network(mgr):
    NetworkFromConfig, _ = import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        return NetworkFromConfig(mgr)
