"""lmms-eval plugin package: ``LMMS_EVAL_PLUGINS=videoitg_amd.lmms_plugin`` makes lmms-eval pick up
``videoitg_amd.lmms_plugin.models.AVAILABLE_MODELS`` (/root/reference/lmms_eval/models/__init__.py:84-89)."""
