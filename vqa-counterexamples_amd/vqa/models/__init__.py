"""placeholder, filled below"""
