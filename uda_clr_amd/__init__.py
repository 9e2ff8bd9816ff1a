"""uda_clr_amd - MI355X-native implementation of the UDA_CLR per-step training hot path.

Sub-packages mirror the reference's import surface (``networks``, ``train_process``, ``utils``,
``dataloaders``); ``uda_clr_amd.dropin.install()`` publishes them under those top-level names so
``train_use_fix_initial.py`` runs unchanged.  All device work goes through the C-ABI library
``libuda_clr_hip.so`` (``include/uda_clr_hip.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"
