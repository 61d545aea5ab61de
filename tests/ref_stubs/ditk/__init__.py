"""ditk: the reference only uses ``from ditk import logging``"""
import logging  # noqa: F401
