def register(*args, **kwargs):
    pass
