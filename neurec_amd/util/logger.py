"""Logger — stdout + file, flushed after every message (util/logger.py:10-70)."""
import logging
import os
import sys


class Logger(object):
    def __init__(self, filename):
        folder = os.path.dirname(filename)
        if folder and not os.path.exists(folder):
            os.makedirs(folder)
        self.logger = logging.getLogger(filename)
        self.logger.setLevel(logging.DEBUG)
        fmt = logging.Formatter("%(asctime)s.%(msecs)03d: %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
        for handler in (logging.FileHandler(filename), logging.StreamHandler(sys.stdout)):
            handler.setLevel(logging.DEBUG)
            handler.setFormatter(fmt)
            self.logger.addHandler(handler)

    def _emit(self, level, message):
        self.logger.log(level, message)
        for handler in self.logger.handlers:
            handler.flush()

    def debug(self, message):
        self._emit(logging.DEBUG, message)

    def info(self, message):
        self._emit(logging.INFO, message)

    def warning(self, message):
        self._emit(logging.WARNING, message)

    def error(self, message):
        self._emit(logging.ERROR, message)

    def critical(self, message):
        self._emit(logging.CRITICAL, message)
