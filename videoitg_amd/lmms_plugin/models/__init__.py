# name -> class name inside ``videoitg_amd.lmms_plugin.models.<name>`` (contract of lmms_eval/models/__init__.py:84-89)
AVAILABLE_MODELS = {"videoitg": "VideoITG"}
