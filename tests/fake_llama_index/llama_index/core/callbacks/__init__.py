class CallbackManager:
    """Records (event, payload) pairs: enough to see that `retrieve` ran inside a RETRIEVE event."""

    def __init__(self, handlers=None):
        self.handlers = handlers or []
        self.events = []

    def on_event(self, name, payload):
        self.events.append((name, payload))
