from .runner import ProcessingRunner

__all__ = ["ProcessingRunner"]
