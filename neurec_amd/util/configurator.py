"""Configurator — ini file + command line, with the reference's value coercion.

Behavioural mirror of util/configurator.py:44-157 so that an unchanged
`NeuRec.properties` / `conf/<Model>.properties` pair and `--key=value` overrides
produce the same values:
  * a file with one section is read whatever the section is called, otherwise the
    section `default_section` is required (configurator.py:85-93);
  * command-line values overwrite same-named file keys at read time (:97-99) and keys
    that exist only on the command line are visible too (:124-125);
  * lookup order: library file, then model file, then command line (:120-127);
  * values are `eval`-ed; anything that does not evaluate to
    str/int/float/list/tuple/bool/None stays a string; `true`/`false` in any case map
    to booleans (:129-140).
"""
import os
import sys
from collections import OrderedDict
from configparser import ConfigParser

_PLAIN_TYPES = (str, int, float, list, tuple, bool, type(None))
_PATH_HOSTILE = set('/\\":*?<>|\t')


def _coerce(text):
    try:
        value = eval(text)
        return value if isinstance(value, _PLAIN_TYPES) else text
    except Exception:
        low = text.lower()
        if low == "true":
            return True
        if low == "false":
            return False
        return text


class Configurator(object):
    def __init__(self, config_file, default_section="default", argv=None):
        """argv: command-line tokens (defaults to sys.argv[1:], as the reference reads them)."""
        if not os.path.isfile(config_file):
            raise FileNotFoundError("There is not config file named '%s'!" % config_file)
        self._default_section = default_section
        self.cmd_arg = self._parse_cmd(argv)
        self.lib_arg = self._read_ini(config_file)
        model_file = os.path.join(self.lib_arg["config_dir"],
                                  self.lib_arg["recommender"] + ".properties")
        self.alg_arg = self._read_ini(model_file)

    @staticmethod
    def _parse_cmd(argv):
        if argv is None:
            argv = [] if "ipykernel_launcher" in sys.argv[0] else sys.argv[1:]
        parsed = OrderedDict()
        for token in argv:
            if not token.startswith("--"):
                raise SyntaxError("Commend arg must start with '--', but '%s' is not!" % token)
            name, value = token[2:].split("=")
            parsed[name] = value
        return parsed

    def _read_ini(self, filename):
        parser = ConfigParser()
        parser.optionxform = str            # keep key case
        parser.read(filename, encoding="utf-8")
        sections = parser.sections()
        if not sections:
            raise ValueError("'%s' is empty!" % filename)
        if len(sections) == 1:
            chosen = sections[0]
        elif self._default_section in sections:
            chosen = self._default_section
        else:
            raise ValueError("'%s' has more than one sections but there is no section named '%s'"
                             % (filename, self._default_section))
        args = OrderedDict(parser[chosen].items())
        for name, value in self.cmd_arg.items():
            if name in args:
                args[name] = value
        return args

    def params_str(self):
        """Summary of the model hyper-parameters used in log file names."""
        joined = "_".join("{}={}".format(k, v) for k, v in self.alg_arg.items() if len(v) < 20)
        cleaned = "".join("_" if ch in _PATH_HOSTILE else ch for ch in joined)
        return "%s_%s" % (self["recommender"], cleaned)

    def __getitem__(self, item):
        if not isinstance(item, str):
            raise TypeError("index must be a str")
        for table in (self.lib_arg, self.alg_arg, self.cmd_arg):
            if item in table:
                return _coerce(table[item])
        raise KeyError("There are not the parameter named '%s'" % item)

    def __getattr__(self, item):
        if item.startswith("_") or item in ("cmd_arg", "lib_arg", "alg_arg"):
            raise AttributeError(item)
        return self[item]

    def __contains__(self, o):
        return o in self.lib_arg or o in self.alg_arg or o in self.cmd_arg

    def __str__(self):
        lib = "\n".join("{}={}".format(k, v) for k, v in self.lib_arg.items())
        alg = "\n".join("{}={}".format(k, v) for k, v in self.alg_arg.items())
        return "\n\nNeuRec hyperparameters:\n%s\n\n%s's hyperparameters:\n%s\n" % (
            lib, self["recommender"], alg)

    __repr__ = __str__
