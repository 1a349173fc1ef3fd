from .proxy_evaluator import ProxyEvaluator
