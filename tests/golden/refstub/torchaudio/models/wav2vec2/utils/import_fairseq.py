def _convert_state_dict(sd):
    return sd
