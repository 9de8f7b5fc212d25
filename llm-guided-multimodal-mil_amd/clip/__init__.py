from .model import CLIPText, build_text_model      # noqa: F401
